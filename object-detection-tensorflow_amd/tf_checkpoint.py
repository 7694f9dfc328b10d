"""TensorFlow checkpoint files without TensorFlow: what `tf.train.Saver` / `NewCheckpointReader` do for the reference.

Reference call sites this stands in for:
  * SSD300.py:31  `wrap.NewCheckpointReader(config['pretraining_weight'])` + `:195-300 reader.get_tensor(name)`
    (slim's vgg_16.ckpt -- a V1 "tensor slice" checkpoint) .............................. `CheckpointReader`
  * SSD300.py:464-466, :490-504  `tf.train.Saver().save / .restore` (V2 "tensor bundle") ... `write_bundle`, `CheckpointReader`
The file formats are TensorFlow's, restated from its sources (no TensorFlow in this environment to produce vectors:
PARITY UNPINNED; the writer and reader check each other and every block / tensor CRC):
  * both formats sit on TensorFlow's copy of the LevelDB table (tensorflow/core/lib/io/table*, format.cc): data blocks of
    prefix-compressed entries with restart points, a 1-byte compression tag + masked CRC32C behind every block, an
    index block of block handles, a 48-byte footer ending in the magic 0xdb4775248b80fb57;
  * V2 / tensor bundle (tensorflow/core/util/tensor_bundle): `<prefix>.index` is such a table, key "" -> BundleHeaderProto,
    key <tensor name> -> BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}; the bytes are in
    `<prefix>.data-0000S-of-0000N` at [offset, offset + size), little endian, row major;
  * V1 / tensor slices (tensorflow/core/util/saved_tensor_slice.proto, tensor_slice_writer.cc): ONE table file, key "" ->
    SavedTensorSlices{meta}, every other key -> SavedTensorSlices{data = SavedSlice{name, slice, TensorProto}}.
Protocol buffers are decoded by hand (wire format only; the field numbers are quoted where used).
"""
from __future__ import annotations

import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
_MASK_DELTA = 0xa282ead8

# tensorflow/core/framework/types.proto
DT_FLOAT, DT_DOUBLE, DT_INT32, DT_UINT8, DT_INT16, DT_INT8, DT_STRING, DT_INT64, DT_BOOL, DT_BFLOAT16, DT_HALF = 1, 2, 3, 4, 5, 6, 7, 9, 10, 14, 19
_NP_OF_DT = {DT_FLOAT: np.float32, DT_DOUBLE: np.float64, DT_INT32: np.int32, DT_UINT8: np.uint8, DT_INT16: np.int16,
             DT_INT8: np.int8, DT_INT64: np.int64, DT_BOOL: np.bool_, DT_HALF: np.float16, DT_BFLOAT16: np.uint16}
_DT_OF_NP = {np.dtype(v): k for k, v in _NP_OF_DT.items() if k != DT_BFLOAT16}


# --------------------------------------------------------------------------------------------------- CRC32C (Castagnoli)
def _make_table():
    tab = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
        tab.append(c)
    return tab


_CRC_TABLE = _make_table()
_native_crc = None


def _load_native_crc():
    """libodtk exports a host-side slice-by-8 CRC32C (odtk_crc32c); hundreds of MB of weights need it"""
    global _native_crc
    if _native_crc is None:
        try:
            from . import _lib
            fn = _lib.load().odtk_crc32c
            _native_crc = fn
        except Exception:                                  # noqa: BLE001 -- reading small files works without the library
            _native_crc = False
    return _native_crc


def crc32c(data, crc: int = 0) -> int:
    mv = memoryview(data).cast('B') if not isinstance(data, (bytes, bytearray)) else data
    n = len(mv)
    if n >= 4096 and _load_native_crc():
        import ctypes as C
        buf = np.frombuffer(mv, dtype=np.uint8)
        return int(_native_crc(C.c_void_p(buf.ctypes.data), C.c_longlong(n), C.c_uint(crc))) & 0xffffffff
    c = crc ^ 0xffffffff
    tab = _CRC_TABLE
    for b in bytes(mv):
        c = tab[(c ^ b) & 0xff] ^ (c >> 8)
    return c ^ 0xffffffff


def mask_crc(c: int) -> int:
    """crc32c::Mask: rotate right by 15 and add a constant (CRCs of data that embeds CRCs)"""
    return (((c >> 15) | (c << 17)) + _MASK_DELTA) & 0xffffffff


# --------------------------------------------------------------------------------------------------- varints / protobuf wire
def _put_varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while n >= 0x80:
        out.append((n & 0x7f) | 0x80)
        n >>= 7
    out.append(n)
    return bytes(out)


def _get_varint(buf, pos: int):
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7f) << shift
        if b < 0x80:
            return val, pos
        shift += 7


def _pb_parse(buf):
    """[(field, wire type, value)]: varint -> int, fixed32/64 -> int, length-delimited -> bytes"""
    out, pos, n = [], 0, len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v = bytes(buf[pos:pos + ln]); pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]; pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        out.append((f, wt, v))
    return out


def _pb_field(f: int, wt: int, payload: bytes) -> bytes:
    return _put_varint((f << 3) | wt) + payload


def _pb_bytes(f: int, b: bytes) -> bytes:
    return _pb_field(f, 2, _put_varint(len(b)) + b)


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf) -> tuple:
    """TensorShapeProto: repeated Dim dim = 2 {int64 size = 1}"""
    dims = []
    for f, _, v in _pb_parse(buf):
        if f == 2:
            size = 0
            for g, _, w in _pb_parse(v):
                if g == 1:
                    size = _signed(w)
            dims.append(size)
    return tuple(dims)


def _encode_shape(shape) -> bytes:
    return b''.join(_pb_bytes(2, _pb_field(1, 0, _put_varint(int(d)))) for d in shape)


# --------------------------------------------------------------------------------------------------- snappy (raw format)
def _snappy_uncompress(src: bytes) -> bytes:
    n, pos = _get_varint(src, 0)
    out = bytearray()
    while pos < len(src):
        tag = src[pos]; pos += 1
        kind = tag & 3
        if kind == 0:                                   # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(src[pos:pos + nb], 'little'); pos += nb
            ln += 1
            out += src[pos:pos + ln]; pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | src[pos]; pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 2], 'little'); pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(src[pos:pos + 4], 'little'); pos += 4
        for _ in range(ln):                             # may overlap its own output
            out.append(out[-off])
    if len(out) != n:
        raise ValueError('snappy: length mismatch')
    return bytes(out)


# --------------------------------------------------------------------------------------------------- the table
class _Table:
    """read side of tensorflow/core/lib/io/table: yields (key, value) in key order"""

    def __init__(self, data: bytes, verify: bool = True):
        if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != TABLE_MAGIC:
            raise ValueError('not a TensorFlow table file (bad magic)')
        self.data, self.verify = data, verify
        foot = data[-48:]
        _, p = _get_varint(foot, 0); _, p = _get_varint(foot, p)          # metaindex handle (unused)
        self.index_off, p = _get_varint(foot, p)
        self.index_size, p = _get_varint(foot, p)

    def _block(self, off: int, size: int) -> bytes:
        raw = self.data[off:off + size]
        kind = self.data[off + size]
        if self.verify:
            want = struct.unpack_from('<I', self.data, off + size + 1)[0]
            if mask_crc(crc32c(self.data[off:off + size + 1])) != want:
                raise ValueError(f'table block at {off}: checksum mismatch')
        if kind == 0:
            return raw
        if kind == 1:
            return _snappy_uncompress(raw)
        raise ValueError(f'table block at {off}: unknown compression {kind}')

    @staticmethod
    def _entries(block: bytes):
        nrestart = struct.unpack_from('<I', block, len(block) - 4)[0]
        end = len(block) - 4 - 4 * nrestart
        pos, key = 0, b''
        while pos < end:
            shared, pos = _get_varint(block, pos)
            nons, pos = _get_varint(block, pos)
            vlen, pos = _get_varint(block, pos)
            key = key[:shared] + block[pos:pos + nons]; pos += nons
            yield key, block[pos:pos + vlen]
            pos += vlen

    def items(self):
        for _, handle in self._entries(self._block(self.index_off, self.index_size)):
            off, p = _get_varint(handle, 0)
            size, _ = _get_varint(handle, p)
            yield from self._entries(self._block(off, size))


def _snappy_literal(data: bytes) -> bytes:
    """a valid (if pointless) snappy stream: the length, then literals of <= 60 bytes -- enough to exercise readers of compressed blocks"""
    out = bytearray(_put_varint(len(data)))
    for i in range(0, len(data), 60):
        chunk = data[i:i + 60]
        out.append((len(chunk) - 1) << 2)
        out += chunk
    return bytes(out)


def _build_table(items, block_size: int = 262144, restart_interval: int = 16, snappy: bool = False) -> bytes:
    """write side: `items` sorted (key, value) pairs, no compression (what BundleWriter asks for); snappy=True marks the blocks
    compressed and stores them as literal-only snappy streams (for tests of the reader)"""
    out = bytearray()
    index = []                                                         # (last key of block, handle)

    def emit(block: bytes):
        off = len(out)
        kind = b'\x01' if snappy else b'\x00'                          # kSnappyCompression / kNoCompression
        if snappy:
            block = _snappy_literal(block)
        out.extend(block)
        out.extend(kind)
        out.extend(struct.pack('<I', mask_crc(crc32c(block + kind))))
        return _put_varint(off) + _put_varint(len(block))

    def block_of(entries, interval):
        buf, restarts, prev = bytearray(), [], b''
        for i, (k, v) in enumerate(entries):
            shared = 0
            if i % interval == 0:
                restarts.append(len(buf))
            else:
                m = min(len(prev), len(k))
                while shared < m and prev[shared] == k[shared]:
                    shared += 1
            buf += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
            prev = k
        if not restarts:
            restarts = [0]
        for r in restarts:
            buf += struct.pack('<I', r)
        buf += struct.pack('<I', len(restarts))
        return bytes(buf)

    cur, cur_bytes, last = [], 0, None
    for k, v in items:
        if last is not None and not last < k:
            raise ValueError('table keys must be strictly increasing')
        last = k
        cur.append((k, v)); cur_bytes += len(k) + len(v) + 8
        if cur_bytes >= block_size:
            index.append((cur[-1][0], emit(block_of(cur, restart_interval))))
            cur, cur_bytes = [], 0
    if cur:
        index.append((cur[-1][0], emit(block_of(cur, restart_interval))))
    meta = emit(block_of([], 1))
    idx = emit(block_of(index, 1))
    foot = meta + idx
    out.extend(foot + b'\x00' * (40 - len(foot)) + struct.pack('<Q', TABLE_MAGIC))
    return bytes(out)


# --------------------------------------------------------------------------------------------------- reader
class CheckpointReader:
    """`tf.train.NewCheckpointReader(path)`: has_tensor / get_tensor / get_variable_to_shape_map /
    get_variable_to_dtype_map, for V2 prefixes (`model.ckpt-1000`) and V1 files (`vgg_16.ckpt`)."""

    def __init__(self, path: str, verify: bool = True):
        self.path, self.verify = str(path), verify
        self._entries = {}                # name -> dict(dtype, shape, ...)
        if os.path.exists(self.path + '.index'):
            self.version = 2
            self._read_bundle_index()
        elif os.path.isfile(self.path):
            self.version = 1
            self._read_slices()
        else:
            raise FileNotFoundError(f'no TensorFlow checkpoint at {self.path!r} (neither {self.path}.index nor the file itself)')

    # ---- V2
    def _read_bundle_index(self):
        with open(self.path + '.index', 'rb') as f:
            table = _Table(f.read(), self.verify)
        self.num_shards = 1
        for key, val in table.items():
            if key == b'':                                           # BundleHeaderProto: num_shards = 1, endianness = 2, version = 3
                for f_, _, v in _pb_parse(val):
                    if f_ == 1:
                        self.num_shards = v
                    elif f_ == 2 and v != 0:
                        raise ValueError('big-endian tensor bundles are not supported')
                continue
            e = dict(dtype=0, shape=(), shard=0, offset=0, size=0, crc=None, slices=False)
            for f_, _, v in _pb_parse(val):                          # BundleEntryProto
                if f_ == 1:
                    e['dtype'] = v
                elif f_ == 2:
                    e['shape'] = _parse_shape(v)
                elif f_ == 3:
                    e['shard'] = v
                elif f_ == 4:
                    e['offset'] = v
                elif f_ == 5:
                    e['size'] = v
                elif f_ == 6:
                    e['crc'] = v
                elif f_ == 7:
                    e['slices'] = True                               # partitioned variable: the slices are separate entries
            self._entries[key.decode()] = e

    def _bundle_tensor(self, name):
        e = self._entries[name]
        if e['slices']:
            raise NotImplementedError(f'{name}: partitioned variables are not supported')
        if e['dtype'] not in _NP_OF_DT:
            raise NotImplementedError(f'{name}: dtype enum {e["dtype"]} is not supported')
        fn = f'{self.path}.data-{e["shard"]:05d}-of-{self.num_shards:05d}'
        with open(fn, 'rb') as f:
            f.seek(e['offset'])
            raw = f.read(e['size'])
        if len(raw) != e['size']:
            raise ValueError(f'{name}: {fn} is truncated')
        if self.verify and e['crc'] is not None and mask_crc(crc32c(raw)) != e['crc']:
            raise ValueError(f'{name}: tensor checksum mismatch in {fn}')
        return np.frombuffer(raw, dtype=_NP_OF_DT[e['dtype']]).reshape(e['shape']).copy()

    # ---- V1
    def _read_slices(self):
        with open(self.path, 'rb') as f:
            table = _Table(f.read(), self.verify)
        self._slices = {}
        for key, val in table.items():
            top = _pb_parse(val)                                      # SavedTensorSlices: meta = 1, data = 2
            if key == b'':
                for f_, _, v in top:
                    if f_ != 1:
                        continue
                    for g, _, w in _pb_parse(v):                      # SavedTensorSliceMeta: repeated SavedSliceMeta tensor = 1
                        if g != 1:
                            continue
                        name, shape, dtype = '', (), 0
                        for h, _, x in _pb_parse(w):                  # SavedSliceMeta: name = 1, shape = 2, type = 3, slice = 4
                            if h == 1:
                                name = x.decode()
                            elif h == 2:
                                shape = _parse_shape(x)
                            elif h == 3:
                                dtype = x
                        self._entries[name] = dict(dtype=dtype, shape=shape)
                continue
            for f_, _, v in top:
                if f_ != 2:
                    continue
                name, extents, tensor = '', [], b''
                for g, _, w in _pb_parse(v):                          # SavedSlice: name = 1, slice = 2, data = 3
                    if g == 1:
                        name = w.decode()
                    elif g == 2:
                        for h, _, x in _pb_parse(w):                  # TensorSliceProto: repeated Extent extent = 1 {start = 1, length = 2}
                            if h == 1:
                                start, length = 0, None
                                for i, _, y in _pb_parse(x):
                                    if i == 1:
                                        start = _signed(y)
                                    elif i == 2:
                                        length = _signed(y)
                                extents.append((start, length))
                    elif g == 3:
                        tensor = w
                self._slices.setdefault(name, []).append((extents, tensor))

    @staticmethod
    def _tensor_proto_values(buf, dtype):
        """TensorProto: dtype = 1, tensor_content = 4, half_val = 13, float_val = 5, double_val = 6, int_val = 7,
        int64_val = 10, bool_val = 11 (repeated scalars arrive packed or one by one)"""
        np_dt = _NP_OF_DT[dtype]
        field = {DT_FLOAT: 5, DT_DOUBLE: 6, DT_INT32: 7, DT_UINT8: 7, DT_INT16: 7, DT_INT8: 7, DT_INT64: 10, DT_BOOL: 11, DT_HALF: 13,
                 DT_BFLOAT16: 13}[dtype]
        chunks = []
        for f_, wt, v in _pb_parse(buf):
            if f_ == 4 and wt == 2 and len(v):
                return np.frombuffer(v, dtype=np_dt)
            if f_ != field:
                continue
            if wt == 2:                                               # packed
                if dtype == DT_FLOAT:
                    chunks.append(np.frombuffer(v, dtype='<f4'))
                elif dtype == DT_DOUBLE:
                    chunks.append(np.frombuffer(v, dtype='<f8'))
                else:
                    vals, pos = [], 0
                    while pos < len(v):
                        x, pos = _get_varint(v, pos)
                        vals.append(_signed(x))
                    chunks.append(np.asarray(vals, dtype=np.int64))
            elif wt == 5:
                chunks.append(np.asarray([struct.unpack('<f', struct.pack('<I', v))[0]], dtype=np.float32))
            elif wt == 1:
                chunks.append(np.asarray([struct.unpack('<d', struct.pack('<Q', v))[0]], dtype=np.float64))
            else:
                chunks.append(np.asarray([_signed(v)], dtype=np.int64))
        flat = np.concatenate(chunks) if chunks else np.zeros(0, np_dt)
        if dtype in (DT_HALF, DT_BFLOAT16):
            return flat.astype(np.uint16).view(np_dt)
        return flat.astype(np_dt)

    def _slice_tensor(self, name):
        e = self._entries[name]
        if e['dtype'] not in _NP_OF_DT:
            raise NotImplementedError(f'{name}: dtype enum {e["dtype"]} is not supported')
        out = np.zeros(e['shape'], dtype=_NP_OF_DT[e['dtype']])
        for extents, tensor in self._slices.get(name, []):
            index, shape = [], []
            for d, dim in enumerate(e['shape']):
                start, length = extents[d] if d < len(extents) else (0, None)
                length = dim - start if length is None else length
                index.append(slice(start, start + length)); shape.append(length)
            out[tuple(index)] = self._tensor_proto_values(tensor, e['dtype']).reshape(shape)
        return out

    # ---- the NewCheckpointReader surface
    def has_tensor(self, name: str) -> bool:
        return name in self._entries

    def get_tensor(self, name: str) -> np.ndarray:
        if name not in self._entries:
            raise KeyError(f'Key {name} not found in checkpoint')       # tensorflow: NotFoundError with this text
        return self._bundle_tensor(name) if self.version == 2 else self._slice_tensor(name)

    def get_variable_to_shape_map(self) -> dict:
        return {k: list(v['shape']) for k, v in self._entries.items()}

    def get_variable_to_dtype_map(self) -> dict:
        return {k: _NP_OF_DT.get(v['dtype']) for k, v in self._entries.items()}


NewCheckpointReader = CheckpointReader


# --------------------------------------------------------------------------------------------------- writer (V2)
def write_bundle(prefix: str, tensors: dict) -> None:
    """What `Saver.save(sess, prefix)` leaves on disk for a one-shard V2 checkpoint: `<prefix>.index` and
    `<prefix>.data-00000-of-00001` (tensors in name order, as BundleWriter's sorted map finishes them)."""
    prefix = str(prefix)
    names = sorted(tensors, key=lambda s: s.encode())
    items = [(b'', _pb_field(1, 0, _put_varint(1)) + _pb_bytes(3, _pb_field(1, 0, _put_varint(1))))]   # num_shards = 1, version.producer = 1
    offset = 0
    tmp = f'{prefix}.data-00000-of-00001.tmp'
    with open(tmp, 'wb') as f:
        for n in names:
            a = np.asarray(tensors[n])                         # (ascontiguousarray would turn a scalar into shape [1])
            if a.dtype not in _DT_OF_NP:
                raise TypeError(f'{n}: dtype {a.dtype} has no TensorFlow counterpart here')
            raw = a.tobytes()
            f.write(raw)
            entry = _pb_field(1, 0, _put_varint(_DT_OF_NP[a.dtype])) + _pb_bytes(2, _encode_shape(a.shape))
            if offset:
                entry += _pb_field(4, 0, _put_varint(offset))
            entry += _pb_field(5, 0, _put_varint(len(raw))) + _pb_field(6, 5, struct.pack('<I', mask_crc(crc32c(raw))))
            items.append((n.encode(), entry))
            offset += len(raw)
    os.replace(tmp, f'{prefix}.data-00000-of-00001')
    with open(prefix + '.index.tmp', 'wb') as f:
        f.write(_build_table(items))
    os.replace(prefix + '.index.tmp', prefix + '.index')


def update_checkpoint_state(prefix: str) -> None:
    """the `checkpoint` text file Saver keeps next to the files (CheckpointState: model_checkpoint_path, all_model_checkpoint_paths)"""
    d, base = os.path.dirname(prefix) or '.', os.path.basename(prefix)
    with open(os.path.join(d, 'checkpoint'), 'w') as f:
        f.write(f'model_checkpoint_path: "{base}"\nall_model_checkpoint_paths: "{base}"\n')


def latest_checkpoint(directory: str):
    """tf.train.latest_checkpoint"""
    fn = os.path.join(directory, 'checkpoint')
    if not os.path.exists(fn):
        return None
    for line in open(fn):
        if line.startswith('model_checkpoint_path:'):
            p = line.split(':', 1)[1].strip().strip('"')
            return p if os.path.isabs(p) else os.path.join(directory, p)
    return None


def main(argv=None):
    """python -m odtk.tf_checkpoint <checkpoint prefix or V1 file> [tensor name]: what inspect_checkpoint prints"""
    import sys
    argv = sys.argv[1:] if argv is None else argv
    if not argv:
        print(main.__doc__)
        return 2
    r = CheckpointReader(argv[0])
    if len(argv) > 1:
        a = r.get_tensor(argv[1])
        print(argv[1], a.dtype, list(a.shape))
        print(a)
        return 0
    shapes, dtypes = r.get_variable_to_shape_map(), r.get_variable_to_dtype_map()
    total = 0
    for k in sorted(shapes):
        n = int(np.prod(shapes[k])) if shapes[k] else 1
        total += n
        print(f'{k}  {getattr(dtypes[k], "__name__", dtypes[k])}  {shapes[k]}')
    print(f'# {len(shapes)} tensors, {total} elements, format V{r.version}')
    return 0


if __name__ == '__main__':
    raise SystemExit(main())
