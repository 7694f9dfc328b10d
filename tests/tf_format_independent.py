"""An INDEPENDENT encoder of TensorFlow's two checkpoint formats, written from the published format descriptions with no code shared with
object-detection-tensorflow_amd/tf_checkpoint.py (own CRC32C, own varints / protobuf wire encoder, own LevelDB-style table builder with
different block sizes and restart intervals than the product's writer).  tests/test_tf_checkpoint_cpu.py feeds its files to the product's
reader: the reader has then decoded bytes that neither it nor its sibling writer produced.

Formats (TensorFlow source tree, from memory; field numbers are the .proto's):
  table file ........ tensorflow/core/lib/io/table_format.txt + block_builder.cc: data blocks of prefix-compressed entries
                      [varint shared][varint non_shared][varint value_len][key delta][value] + uint32 restart offsets + uint32 count; every block
                      is followed by a 1-byte type (0 = raw) and the MASKED crc32c of block + type; index block maps a separator key to a
                      BlockHandle (varint offset, varint size); footer = metaindex handle + index handle, zero-padded to 40 bytes + magic
                      0xdb4775248b80fb57 little-endian.
  V2 bundle ......... tensor_bundle.proto: key '' -> BundleHeaderProto {1: num_shards, 2: endianness (0 = little), 3: VersionDef {1: producer}};
                      key <name> -> BundleEntryProto {1: dtype, 2: TensorShapeProto {2: Dim {1: size}}, 3: shard_id, 4: offset, 5: size,
                      6: fixed32 masked crc32c of the tensor bytes}; `<prefix>.data-00000-of-00001` = the tensors' bytes in key order.
  V1 slices ......... saved_tensor_slice.proto: key '' -> SavedTensorSlices {1: SavedTensorSliceMeta {1: SavedSliceMeta {1: name, 2: shape,
                      3: type, 4: TensorSliceProto {1: Extent {1: start, 2: length}}}, 2: VersionDef}}; one entry per slice ->
                      SavedTensorSlices {2: SavedSlice {1: name, 2: slice, 3: TensorProto {1: dtype, 2: shape, 5: packed float_val | 7: int_val |
                      10: int64_val}}}; keys = OrderedCode(0, name, dims...) -- only their ORDER matters to a reader.
"""
import struct

import numpy as np

DT_FLOAT, DT_INT32, DT_INT64 = 1, 3, 9


def crc32c_bitwise(data: bytes) -> int:
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), bit by bit -- RFC 3720 appendix B.4"""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ (0x82F63B78 if crc & 1 else 0)
    return crc ^ 0xFFFFFFFF


def crc32c_fast(data: bytes) -> int:
    """the same function by a 256-entry table built from the bitwise form (large tensors)"""
    tab = getattr(crc32c_fast, '_t', None)
    if tab is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            tab.append(c)
        crc32c_fast._t = tab
    a = np.frombuffer(data, dtype=np.uint8)
    crc = 0xFFFFFFFF
    for b in a.tolist():
        crc = tab[(crc ^ b) & 0xFF] ^ (crc >> 8)
    return crc ^ 0xFFFFFFFF


def masked(crc: int) -> int:
    """crc32c::Mask: rotate right by 15, add 0xa282ead8"""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def varint(n: int) -> bytes:
    out = bytearray()
    n &= (1 << 64) - 1
    while True:
        low = n & 0x7F
        n >>= 7
        out.append(low | (0x80 if n else 0))
        if not n:
            return bytes(out)


def field_varint(num: int, v: int) -> bytes:
    return varint(num << 3 | 0) + varint(v)


def field_bytes(num: int, b: bytes) -> bytes:
    return varint(num << 3 | 2) + varint(len(b)) + b


def field_fixed32(num: int, v: int) -> bytes:
    return varint(num << 3 | 5) + struct.pack('<I', v)


def shape_proto(shape) -> bytes:
    return b''.join(field_bytes(2, field_varint(1, int(d))) for d in shape)


def build_table(items, block_bytes=700, restart_every=3) -> bytes:
    """items: sorted [(key, value)] -> the bytes of a table file"""
    out = bytearray()
    index_entries = []

    def emit_block(payload: bytes):
        off = len(out)
        out.extend(payload)
        out.append(0)                                                        # kNoCompression
        out.extend(struct.pack('<I', masked(crc32c_fast(payload + b'\x00'))))
        return varint(off) + varint(len(payload))

    def block_of(entries):
        body, restarts, prev = bytearray(), [], b''
        for i, (k, v) in enumerate(entries):
            shared = 0
            if i % restart_every == 0:
                restarts.append(len(body))
            else:
                while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                    shared += 1
            body += varint(shared) + varint(len(k) - shared) + varint(len(v)) + k[shared:] + v
            prev = k
        for r in restarts:
            body += struct.pack('<I', r)
        body += struct.pack('<I', len(restarts))
        return bytes(body)
    cur, size = [], 0
    for k, v in items:
        cur.append((k, v)); size += len(k) + len(v)
        if size >= block_bytes:
            index_entries.append((cur[-1][0], emit_block(block_of(cur))))    # separator = the block's last key (any key >= it and < the next works)
            cur, size = [], 0
    if cur:
        index_entries.append((cur[-1][0], emit_block(block_of(cur))))
    meta = emit_block(struct.pack('<I', 0) + struct.pack('<I', 1))            # empty metaindex block: one restart at 0
    idx_handle = emit_block(block_of(index_entries))
    footer = meta + idx_handle
    out.extend(footer + bytes(40 - len(footer)) + struct.pack('<Q', 0xDB4775248B80FB57))
    return bytes(out)


def _dtype(a):
    return {np.dtype('float32'): DT_FLOAT, np.dtype('int32'): DT_INT32, np.dtype('int64'): DT_INT64}[a.dtype]


def write_v2_bundle(prefix: str, tensors: dict):
    names = sorted(tensors, key=lambda s: s.encode())
    items = [(b'', field_varint(1, 1) + field_varint(2, 0) + field_bytes(3, field_varint(1, 1)))]
    data = bytearray()
    for n in names:
        a = np.asarray(tensors[n])                                        # (np.ascontiguousarray would promote a scalar to shape [1])
        raw = a.astype(a.dtype.newbyteorder('<')).tobytes(order='C')
        entry = field_varint(1, _dtype(a)) + field_bytes(2, shape_proto(a.shape)) + field_varint(4, len(data)) + field_varint(5, len(raw)) \
            + field_fixed32(6, masked(crc32c_fast(raw)))                     # shard_id 0 is the proto default: left out, as protobuf does
        if a.ndim == 0:
            entry = field_varint(1, _dtype(a)) + field_bytes(2, b'') + field_varint(4, len(data)) + field_varint(5, len(raw)) + field_fixed32(6, masked(crc32c_fast(raw)))
        items.append((n.encode(), entry))
        data += raw
    open(prefix + '.index', 'wb').write(build_table(items))
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))


def _ordered_key(name: str, ndim: int, part: int) -> bytes:
    """a key that sorts like TensorFlow's EncodeTensorNameSlice (0, escaped name, then the slice): readers only rely on the order"""
    esc = name.encode().replace(b'\x00', b'\x00\xff').replace(b'\xff', b'\xff\x00') + b'\x00\x01'
    return b'\x00' + esc + bytes([ndim]) + bytes([part])


def write_v1_slices(path: str, tensors: dict, split_first_dim=()):
    metas, entries = b'', []
    for name, a in tensors.items():
        a = np.asarray(a)
        whole = b''.join(field_bytes(1, b'') for _ in a.shape)              # an Extent without `length`: the full dimension
        parts = [(whole, a)]
        if name in split_first_dim:
            h = a.shape[0] // 2
            rest = b''.join(field_bytes(1, b'') for _ in a.shape[1:])
            parts = [(field_bytes(1, field_varint(1, 0) + field_varint(2, h)) + rest, a[:h]),
                     (field_bytes(1, field_varint(1, h) + field_varint(2, a.shape[0] - h)) + rest, a[h:])]
        metas += field_bytes(1, field_bytes(1, name.encode()) + field_bytes(2, shape_proto(a.shape)) + field_varint(3, _dtype(a))
                             + b''.join(field_bytes(4, sl) for sl, _ in parts))
        for i, (sl, arr) in enumerate(parts):
            flat = np.asarray(arr).reshape(-1)
            if arr.dtype == np.float32:
                vals = field_bytes(5, flat.astype('<f4').tobytes())          # packed repeated float
            elif arr.dtype == np.int32:
                vals = field_bytes(7, b''.join(varint(int(x)) for x in flat))
            else:
                vals = field_bytes(10, b''.join(varint(int(x)) for x in flat))
            tensor = field_varint(1, _dtype(arr)) + field_bytes(2, shape_proto(arr.shape)) + vals
            entries.append((_ordered_key(name, arr.ndim, i), field_bytes(2, field_bytes(1, name.encode()) + field_bytes(2, sl) + field_bytes(3, tensor))))
    items = [(b'', field_bytes(1, metas + field_bytes(2, field_varint(1, 1))))] + sorted(entries)
    open(path, 'wb').write(build_table(items, block_bytes=4096, restart_every=5))


def vgg16_slim_tensors(seed=0, fc=False):
    """the variables of slim's vgg_16.ckpt (the file SSD300.py:31 initialises from), real names and shapes for the 13 convolutions and
    fc8 / mean_rgb / global_step; fc6 / fc7 (411 MB + 67 MB in the real file) only on request"""
    g = np.random.default_rng(seed)
    t = {}
    cfg = [('conv1', [(3, 64), (64, 64)]), ('conv2', [(64, 128), (128, 128)]), ('conv3', [(128, 256), (256, 256), (256, 256)]),
           ('conv4', [(256, 512), (512, 512), (512, 512)]), ('conv5', [(512, 512), (512, 512), (512, 512)])]
    for blk, convs in cfg:
        for i, (ci, co) in enumerate(convs):
            t[f'vgg_16/{blk}/{blk}_{i + 1}/weights'] = (g.standard_normal((3, 3, ci, co)) * 0.05).astype(np.float32)
            t[f'vgg_16/{blk}/{blk}_{i + 1}/biases'] = g.standard_normal(co).astype(np.float32)
    if fc:
        t['vgg_16/fc6/weights'] = (g.standard_normal((7, 7, 512, 4096)) * 0.01).astype(np.float32)
        t['vgg_16/fc6/biases'] = g.standard_normal(4096).astype(np.float32)
        t['vgg_16/fc7/weights'] = (g.standard_normal((1, 1, 4096, 4096)) * 0.01).astype(np.float32)
        t['vgg_16/fc7/biases'] = g.standard_normal(4096).astype(np.float32)
    t['vgg_16/fc8/weights'] = (g.standard_normal((1, 1, 4096, 1000)) * 0.01).astype(np.float32)
    t['vgg_16/fc8/biases'] = g.standard_normal(1000).astype(np.float32)
    t['vgg_16/mean_rgb'] = np.asarray([123.68, 116.78, 103.94], np.float32)
    t['global_step'] = np.asarray(0, np.int64)
    return t
