"""odtk.tf_checkpoint (the stand-in for tf.train.Saver / NewCheckpointReader, SSD300.py:31, :464-504) on the CPU.
No TensorFlow here, so the anchors are published ones: the RFC 3720 CRC32C vectors, a table assembled BY HAND from the
LevelDB table-format description, a snappy stream written out by its format description; then writer <-> reader."""
import contextlib
import os
import struct

import numpy as np
import pytest

from odtk import tf_checkpoint as T


def test_crc32c_known_answers():
    assert T.crc32c(b'123456789') == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA                    # RFC 3720 B.4
    assert T.crc32c(b'\xff' * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(bytes(range(31, -1, -1))) == 0x113FDB5C
    big = (bytes(range(256)) * 64)[:16001]                       # native slice-by-8 path (>= 4096 bytes) against the byte loop
    slow = 0xffffffff
    for b in big:
        slow = T._CRC_TABLE[(slow ^ b) & 0xff] ^ (slow >> 8)
    assert T.crc32c(big) == slow ^ 0xffffffff
    assert T.crc32c(big[5000:], T.crc32c(big[:5000])) == T.crc32c(big)      # continuation
    assert T.mask_crc(0) == 0xa282ead8 and T.mask_crc(0xE3069283) == ((0xE3069283 >> 15 | 0xE3069283 << 17) + 0xa282ead8) & 0xffffffff


def _hand_table():
    """two entries in one data block, written straight from the LevelDB table_format / block layout description"""
    def entry(shared, key_delta, value):
        return bytes([shared, len(key_delta), len(value)]) + key_delta + value
    data = entry(0, b'apple', b'1') + entry(2, b'ricot', b'22')             # 'apricot' shares 'ap'
    data += struct.pack('<I', 0) + struct.pack('<I', 1)                      # one restart at 0
    out = bytearray()

    def put(block):
        off = len(out)
        out.extend(block + b'\x00' + struct.pack('<I', T.mask_crc(T.crc32c(block + b'\x00'))))
        return bytes([off, len(block)])                                      # both < 128: one-byte varints
    h_data = put(data)
    h_meta = put(struct.pack('<I', 0) + struct.pack('<I', 1))
    index = entry(0, b'b', h_data) + struct.pack('<I', 0) + struct.pack('<I', 1)
    h_index = put(index)
    foot = h_meta + h_index
    out.extend(foot + bytes(40 - len(foot)) + struct.pack('<Q', 0xdb4775248b80fb57))
    return bytes(out)


def test_table_reader_on_hand_assembled_table_and_writer_roundtrip():
    assert list(T._Table(_hand_table()).items()) == [(b'apple', b'1'), (b'apricot', b'22')]
    bad = bytearray(_hand_table()); bad[3] ^= 1
    with pytest.raises(ValueError, match='checksum'):
        list(T._Table(bytes(bad)).items())
    with pytest.raises(ValueError, match='magic'):
        T._Table(b'\x00' * 64)
    items = [(f'var/{i:04d}/kernel'.encode(), os.urandom(i % 50)) for i in range(300)]
    for bs in (64, 4096, 262144):
        assert list(T._Table(T._build_table(items, block_size=bs)).items()) == items
    with pytest.raises(ValueError, match='increasing'):
        T._build_table([(b'b', b''), (b'a', b'')])


def test_snappy_stream_by_format_description():
    # "hello hello hello!" = literal 'hello ' + copy(offset 6, len 11 -> overlapping) + literal '!'
    src = bytes([18]) + bytes([(6 - 1) << 2]) + b'hello ' + bytes([((11 - 4) << 2) | 1 | (0 << 5), 6]) + bytes([0 << 2]) + b'!'
    assert T._snappy_uncompress(src) == b'hello hello hello!'
    src2 = bytes([10]) + bytes([(4 - 1) << 2]) + b'abcd' + bytes([((6 - 1) << 2) | 2]) + struct.pack('<H', 4)     # 2-byte-offset copy
    assert T._snappy_uncompress(src2) == b'abcdabcdab'


def test_bundle_write_read_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {'feature_extractor/kernel_conv1_1': rng.standard_normal((3, 3, 3, 64)).astype(np.float32),
               'feature_extractor/bias_conv1_1': rng.standard_normal(64).astype(np.float32),
               'global_step': np.asarray(1234, dtype=np.int32),
               'regressor/pred1/kernel': rng.standard_normal((3, 3, 512, 100)).astype(np.float32),     # 1.8 MB: native CRC path
               'misc/half': rng.standard_normal((5, 7)).astype(np.float16), 'misc/i64': np.arange(6, dtype=np.int64).reshape(2, 3),
               'misc/empty': np.zeros((0, 4), np.float32)}
    prefix = str(tmp_path / 'model.ckpt-1234')
    T.write_bundle(prefix, tensors)
    T.update_checkpoint_state(prefix)
    assert sorted(os.listdir(tmp_path)) == ['checkpoint', 'model.ckpt-1234.data-00000-of-00001', 'model.ckpt-1234.index']
    assert T.latest_checkpoint(str(tmp_path)) == prefix
    r = T.NewCheckpointReader(prefix)
    assert r.version == 2 and set(r.get_variable_to_shape_map()) == set(tensors)
    assert r.get_variable_to_shape_map()['global_step'] == [] and r.get_variable_to_dtype_map()['global_step'] == np.int32
    for k, v in tensors.items():
        got = r.get_tensor(k)
        assert got.dtype == v.dtype and got.shape == v.shape and np.array_equal(got, v)
    assert r.has_tensor('global_step') and not r.has_tensor('nope')
    with pytest.raises(KeyError, match='not found in checkpoint'):
        r.get_tensor('nope')
    # the bytes sit in name order and a flipped bit is caught by the tensor CRC
    fn = prefix + '.data-00000-of-00001'
    raw = bytearray(open(fn, 'rb').read())
    assert len(raw) == sum(v.nbytes for v in tensors.values())
    raw[10] ^= 0x40
    open(fn, 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='checksum'):
        T.CheckpointReader(prefix).get_tensor(sorted(tensors, key=str.encode)[0])
    with pytest.raises(FileNotFoundError):
        T.CheckpointReader(str(tmp_path / 'missing'))


def _v1_file(path, tensors, packed=True, split=None):
    """a V1 'tensor slice' checkpoint as tensor_slice_writer.cc lays it out (the test's own encoder)"""
    pb, vi = T._pb_bytes, T._put_varint
    metas, datas = b'', []
    for name, a in tensors.items():
        dt = T._DT_OF_NP[a.dtype]
        full = b''.join(pb(1, b'') for _ in a.shape)                       # Extent without length = the whole dimension
        metas += pb(1, pb(1, name.encode()) + pb(2, T._encode_shape(a.shape)) + T._pb_field(3, 0, vi(dt)) + pb(4, full))
        parts = [(full, a)]
        if split == name:                                                  # two slices along dim 0
            h = a.shape[0] // 2
            ext = lambda s, n: pb(1, T._pb_field(1, 0, vi(s)) + T._pb_field(2, 0, vi(n)))
            rest = b''.join(pb(1, b'') for _ in a.shape[1:])
            parts = [(ext(0, h) + rest, a[:h]), (ext(h, a.shape[0] - h) + rest, a[h:])]
        for i, (sl, arr) in enumerate(parts):
            flat = arr.reshape(-1)
            if a.dtype == np.float32:
                vals = pb(5, flat.astype('<f4').tobytes()) if packed else b''.join(T._pb_field(5, 5, struct.pack('<f', float(x))) for x in flat)
            else:
                vals = pb(7, b''.join(vi(int(x)) for x in flat))
            tp = T._pb_field(1, 0, vi(dt)) + pb(2, T._encode_shape(arr.shape)) + vals
            datas.append((b'\x00' + name.encode() + bytes([i + 1]), pb(2, pb(1, name.encode()) + pb(2, sl) + pb(3, tp))))
    items = [(b'', pb(1, metas))] + sorted(datas)
    open(path, 'wb').write(T._build_table(items, block_size=512))


@pytest.mark.parametrize('packed', [True, False])
def test_v1_slice_checkpoint(tmp_path, packed):
    rng = np.random.default_rng(1)
    tensors = {'vgg_16/conv1/conv1_1/weights': rng.standard_normal((3, 3, 3, 8)).astype(np.float32),
               'vgg_16/conv1/conv1_1/biases': rng.standard_normal(8).astype(np.float32),
               'global_step': np.asarray([7], dtype=np.int32)}
    fn = str(tmp_path / 'vgg_16.ckpt')
    _v1_file(fn, tensors, packed=packed, split='vgg_16/conv1/conv1_1/weights')
    r = T.NewCheckpointReader(fn)
    assert r.version == 1 and r.get_variable_to_shape_map()['vgg_16/conv1/conv1_1/weights'] == [3, 3, 3, 8]
    for k, v in tensors.items():
        assert np.array_equal(r.get_tensor(k), v)


def test_reference_variable_map_matches_reference_graph():
    """odtk.ssd300.reference_variable_map against the variables the reference's own SSD300 class creates
    (tests/golden/ssd300_variables.json, tests/golden/make_golden_variables.py), misspelt names included"""
    import json
    from odtk.ssd300 import reference_variable_map
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ssd300_variables.json')))
    m = reference_variable_map()
    assert set(m) | {'global_step'} == set(want)
    assert 'feature_extractor/kenrel_conv2_1' in m and 'feature_extractor/bias_conv_3_1' in m
    for name, ours in m.items():
        assert want[name]['trainable'] == (not ours.endswith(('.mmean', '.mvar'))), name
        if ours.endswith('.w'):
            assert len(want[name]['shape']) == 4


def test_yolov3_variable_map_matches_reference_graph():
    """odtk.yolov3.reference_variable_map / layer_specs against the variables the reference's own YOLOv3 class creates
    (tests/golden/yolov3_variables.json, tests/golden/make_golden_yolov3_train.py), shapes included"""
    import json
    from odtk.yolov3 import layer_specs, reference_variable_map
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'yolov3_variables.json')))
    m = reference_variable_map()
    assert set(m) | {'global_step'} == set(want) and len(m) == 450
    specs = {s[0]: s for s in layer_specs(20, 3)}
    for name, ours in m.items():
        layer, kind = ours.split('.')
        _, cin, cout, k, _, _ = specs[layer]
        assert want[name]['shape'] == ([k, k, cin, cout] if kind == 'w' else [cout]), name
        assert want[name]['trainable'] == (kind not in ('mmean', 'mvar')), name
    from oracle import yolov3_net_ref as NR
    assert [s[:5] for s in layer_specs(20, 3)] == [s[:5] for s in NR.layer_specs(20, 3)]
    assert [bool(s[5]) for s in layer_specs(20, 3)] == [s[5] is not None for s in NR.layer_specs(20, 3)]


def test_retinanet_variable_map_matches_reference_graph():
    """odtk.retinanet.reference_variable_map against the 733 variables the reference's own RetinaNet class creates
    (tests/golden/retinanet_variables.json, tests/golden/make_golden_retinanet_net.py), shapes and trainable flags included"""
    import json
    from odtk.retinanet import layer_specs, reference_variable_map
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'retinanet_variables.json')))
    m = reference_variable_map()
    assert set(m) | {'global_step'} == set(want) and len(m) == 732
    specs = {s[0]: s for s in layer_specs([3, 4, 6, 3], 16, 21, 9)}
    for name, ours in m.items():
        layer, kind = ours.split('.')
        _, cin, cout, k, _, bnc, _ = specs[layer]
        shape = [k, k, cin, cout] if kind == 'w' else ([cout] if kind == 'b' else [bnc])
        assert want[name]['shape'] == shape, name
        assert want[name]['trainable'] == (kind not in ('mmean', 'mvar')), name


def test_snappy_compressed_tables_and_cli(tmp_path, capsys):
    """tables whose blocks are marked compressed (type 1, snappy) read back like plain ones -- V1 checkpoints written with snappy
    available -- and the inspect-style command line lists a bundle"""
    items = [(f'k{i:05d}'.encode(), os.urandom(1 + i % 97)) for i in range(500)]
    for bs in (128, 4096):
        assert list(T._Table(T._build_table(items, block_size=bs, snappy=True)).items()) == items
    rng = np.random.default_rng(2)
    tensors = {'a/kernel': rng.standard_normal((3, 3, 4, 8)).astype(np.float32), 'global_step': np.asarray(5, np.int32)}
    fn = str(tmp_path / 'v1.ckpt')
    _v1_file(fn, {'a/kernel': tensors['a/kernel']})
    raw = open(fn, 'rb').read()
    entries = list(T._Table(raw).items())
    open(fn, 'wb').write(T._build_table(entries, block_size=256, snappy=True))          # the same V1 content in compressed blocks
    assert np.array_equal(T.CheckpointReader(fn).get_tensor('a/kernel'), tensors['a/kernel'])
    prefix = str(tmp_path / 'm.ckpt-5')
    T.write_bundle(prefix, tensors)
    assert T.main([prefix]) == 0
    out = capsys.readouterr().out
    assert 'a/kernel  float32  [3, 3, 4, 8]' in out and '# 2 tensors, 289 elements, format V2' in out
    assert T.main([prefix, 'global_step']) == 0 and 'int32' in capsys.readouterr().out


def test_fcos_variable_map_and_saver_roundtrip_through_mocked_launches(tmp_path):
    """odtk.fcos.reference_variable_map against the 345 variables of the reference's own FCOS class (tests/golden/fcos_variables.json;
    default layer names numbered per scope, the 11 head layers shared by the five levels), and a tf.train.Saver round trip of the class on the CPU (tests/mock_ops.py)"""
    import json
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mock_ops
    import odtk
    from odtk.fcos import layer_specs, reference_variable_map
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'fcos_variables.json')))
    m = reference_variable_map()
    assert set(m) | {'global_step'} == set(want) and len(m) == 344          # 86 layers x (kernel, bias, gamma, beta): the heads are shared
    specs = {s[0]: s for s in layer_specs(20)}
    for name, ours in m.items():
        layer, kind = ours.split('.')
        _, cin, cout, k, _, gnc, _ = specs[layer]
        assert want[name]['shape'] == ([k, k, cin, cout] if kind == 'w' else ([cout] if kind == 'b' else [gnc])), name
    cfg = {'mode': 'train', 'data_shape': [64, 64, 3], 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
           'batch_size': 1, 'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.45, 'verbose': False, 'device': 'cpu',
           'checkpoint_format': 'tf'}
    prov = {'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    with mock_ops.installed():
        a = odtk.FCOS(dict(cfg, seed=1), prov)
        a.Mom.copy_(torch.randn(a.Mom.shape))
        a.global_step = 7
        path = str(tmp_path / 'ck' / 'fcos.ckpt')
        a.save_weight('latest', path)
        r = T.NewCheckpointReader(path + '-7')
        shapes = r.get_variable_to_shape_map()
        assert all(shapes[n] == v['shape'] for n, v in want.items()) and len(shapes) == 2 * 344 + 1
        b = odtk.FCOS(dict(cfg, seed=2), prov)
        b.load_weight(path + '-7')
        pa, pb = a.export_params(), b.export_params()
        assert all(torch.equal(pa[k], pb[k]) for k in pa) and b.global_step == 7
        for k in ('l0.w', 'l64.gamma', 'l85.b'):
            assert torch.equal(a.get_param(k, a.Mom), b.get_param(k, b.Mom)), k
        c = odtk.FCOS(dict(cfg, seed=3), prov)
        before = c.export_params()
        c.load_pretrained_weight(path + '-7')
        pc = c.export_params()
        assert all(torch.equal(pc[k], pa[k] if int(k[1:].split('.')[0]) < 65 else before[k]) for k in pa)


def test_reader_on_files_of_an_independent_encoder(tmp_path):
    """tests/tf_format_independent.py writes both checkpoint formats with NO code shared with tf_checkpoint.py (own CRC32C -- checked here
    against the RFC 3720 vectors --, own protobuf wire encoder, own table builder with other block sizes / restart intervals, other proto
    field choices: default-valued fields left out).  A whole vgg_16-shaped V1 file (real variable names and shapes of slim's checkpoint,
    74 MB) and a V2 bundle must come back bit for bit through the product's reader -- and the V1 file must initialise SSD300's trunk."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import tf_format_independent as I
    assert I.crc32c_bitwise(bytes(32)) == 0x8A9136AA and I.crc32c_bitwise(b'\xff' * 32) == 0x62A8AB43          # RFC 3720 B.4
    assert I.crc32c_bitwise(bytes(range(32))) == 0x46DD794E and I.crc32c_fast(b'123456789') == I.crc32c_bitwise(b'123456789') == 0xE3069283
    vgg = I.vgg16_slim_tensors(seed=3)
    fn = str(tmp_path / 'vgg_16.ckpt')
    I.write_v1_slices(fn, vgg, split_first_dim=('vgg_16/fc8/weights', 'vgg_16/conv4/conv4_2/biases'))
    r = T.CheckpointReader(fn)
    assert r.version == 1 and set(r.get_variable_to_shape_map()) == set(vgg)
    for name, a in vgg.items():
        got = r.get_tensor(name)
        assert got.dtype == a.dtype and got.shape == a.shape and np.array_equal(got, a), name
    assert r.get_variable_to_shape_map()['vgg_16/conv1/conv1_1/weights'] == [3, 3, 3, 64]
    # V2: a Saver-style bundle incl. a scalar int64 and momentum slots
    g = np.random.default_rng(8)
    tensors = {'feature_extractor/kernel_conv1_1': g.standard_normal((3, 3, 3, 64)).astype(np.float32),
               'feature_extractor/bias_conv1_1': g.standard_normal(64).astype(np.float32),
               'inference/feature_extractor/kernel_conv1_1/Momentum': g.standard_normal((3, 3, 3, 64)).astype(np.float32),
               'regressor/batch_normalization_5/moving_variance': g.random(100).astype(np.float32),
               'regressor/pred6/kernel': g.standard_normal((3, 3, 256, 100)).astype(np.float32),
               'global_step': np.asarray(12345, np.int64), 'some/int32': np.arange(-5, 7, dtype=np.int32).reshape(3, 4)}
    prefix = str(tmp_path / 'model.ckpt-12345')
    I.write_v2_bundle(prefix, tensors)
    r2 = T.CheckpointReader(prefix)
    assert r2.version == 2 and set(r2.get_variable_to_shape_map()) == set(tensors)
    for name, a in tensors.items():
        got = r2.get_tensor(name)
        assert got.dtype == a.dtype and got.shape == a.shape and np.array_equal(got, a), name
    # a flipped data byte must fail the per-tensor checksum
    raw = bytearray(open(prefix + '.data-00000-of-00001', 'rb').read()); raw[100] ^= 0x40
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(raw))
    with pytest.raises(ValueError, match='crc|checksum'):
        [T.CheckpointReader(prefix).get_tensor(n) for n in tensors]
    # the vgg file through the class: SSD300.py:31, :193-299
    import mock_ops
    import odtk
    with mock_ops.installed():
        m = odtk.SSD300({'mode': 'test', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
                         'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': fn, 'verbose': False,
                         'compute_dtype': 'f32', 'device': 'cpu'}, None)
        for n in ('conv1_1', 'conv3_3', 'conv5_3'):
            blk = n.split('_')[0]
            assert np.array_equal(m.get_param(n + '.w').permute(1, 2, 3, 0).numpy(), vgg[f'vgg_16/{blk}/{n}/weights'])
            assert np.array_equal(m.get_param(n + '.b').numpy(), vgg[f'vgg_16/{blk}/{n}/biases'])


@pytest.mark.parametrize("kind", ["refinedet", "pfpnet", "yolov2"])
def test_engine_models_reference_names_and_saver_round_trip(kind, tmp_path):
    """RefineDet320 / PFPNetR / YOLOv2: the rule that derives the reference graph's variable names (default tf.layers names numbered per variable scope,
    the VGG variables as the reference spells them) gives exactly the names of the graph the reference builds on the shim (tests/golden/*_names.json); a
    `checkpoint_format='tf'` save writes tf.train.Saver files (weights HWIO, transposed convs [h, w, out, in], moving statistics, Momentum slots,
    global_step) that a second model restores bit for bit"""
    import json
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import mock_ops
    import odtk
    from odtk.tf_checkpoint import NewCheckpointReader
    if kind == 'yolov2':
        from oracle import yolov2_ref as YR
        cfg = {'mode': 'train', 'is_pretraining': False, 'data_shape': [64, 64, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
               'data_format': 'channels_last', 'batch_size': 1, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'nms_score_threshold': 0.5,
               'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'rescore_confidence': False, 'priors': YR.PRIORS, 'verbose': False, 'compute_dtype': 'f32',
               'device': 'cpu', 'checkpoint_format': 'tf'}
        prov = {'data_shape': [64, 64, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None}
        make = lambda seed: odtk.YOLOv2(dict(cfg, seed=seed), prov)          # noqa: E731
    else:
        cfg = {'mode': 'train', 'input_size': 64, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
               'nms_score_threshold': 0.1, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'pretraining_weight': '', 'verbose': False, 'compute_dtype': 'f32',
               'device': 'cpu', 'checkpoint_format': 'tf'}
        prov = {'data_shape': [64, 64, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None}
        cls = odtk.RefineDet320 if kind == 'refinedet' else odtk.PFPNetR
        make = lambda seed: cls(dict(cfg, seed=seed), prov)                  # noqa: E731
    with mock_ops.installed():
        m = make(1)
        names = m.reference_variable_map()
        want = json.load(open(os.path.join(here, 'golden', f'{kind}_names.json')))
        assert names == want
        g = torch.Generator().manual_seed(5)
        m.Mom.copy_(torch.randn(m.Mom.shape, generator=g) * 1e-3)
        for k in m.sinfo:
            m.stat(k).copy_(torch.rand(m.stat(k).shape, generator=g) + 0.5)
        m.global_step = 37
        prefix = str(tmp_path / 'ck' / kind)
        m.save_weight('latest', prefix)
        reader = NewCheckpointReader(prefix + '-37')
        shapes = reader.get_variable_to_shape_map()
        tf_vars = json.load(open(os.path.join(here, 'golden', f'{kind}_variables.json')))
        assert set(tf_vars) <= set(shapes)                                   # every variable of the reference's graph is in the file ...
        for n, meta in tf_vars.items():                                     # ... with the shape the reference's graph gives it (at this input size)
            assert list(shapes[n]) == meta['shape'], (n, shapes[n], meta['shape'])
        slot = m.MOMENTUM_SLOT_SCOPE + names[m.specs[-1][0] + '.w'] + '/Momentum'
        assert slot in shapes and int(reader.get_tensor('global_step')) == 37
        m2 = make(2)
        assert not torch.equal(m2.P, m.P)
        m2.load_weight(prefix + '-37')
        assert torch.equal(m2.P, m.P) and torch.equal(m2.S, m.S) and m2.global_step == 37
        for k in m.pinfo:                                                    # (the flat momentum buffer was filled including its padding: compare the variables)
            assert torch.equal(m2.get_param(k, m2.Mom), m.get_param(k, m.Mom)), k


def test_centernet_saver_round_trip_through_mocked_launches(tmp_path):
    """CenterNet with `checkpoint_format='tf'`: tf.train.Saver files with every variable of the reference's graph (tests/golden/centernet_variables.json) in the
    reference's shapes, AdamOptimizer's slots and beta-power accumulators; restore is bit-exact; `load_pretrained_weight` takes the backbone only"""
    import json
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import mock_ops
    import odtk
    from odtk.tf_checkpoint import NewCheckpointReader
    cfg = {'mode': 'train', 'input_size': 128, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
           'score_threshold': 0.1, 'top_k_results_output': 10, 'verbose': False, 'compute_dtype': 'f32', 'device': 'cpu', 'checkpoint_format': 'tf'}
    prov = {'data_shape': [128, 128, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    with mock_ops.installed():
        m = odtk.CenterNet(dict(cfg, seed=1), prov)
        g = torch.Generator().manual_seed(6)
        for k in m.pinfo:
            m.param(k, m.M1).copy_(torch.randn(m.param(k, m.M1).shape, generator=g) * 1e-3)
            m.param(k, m.M2).copy_(torch.rand(m.param(k, m.M2).shape, generator=g) * 1e-6)
        for k in m.sinfo:
            m.stat(k).copy_(torch.rand(m.stat(k).shape, generator=g) + 0.5)
        m.global_step = 11
        prefix = str(tmp_path / 'ck' / 'centernet')
        m.save_weight('latest', prefix)
        reader = NewCheckpointReader(prefix + '-11')
        shapes = reader.get_variable_to_shape_map()
        want = json.load(open(os.path.join(here, 'golden', 'centernet_variables.json')))
        for n, meta in want.items():
            assert list(shapes[n]) == meta['shape'], (n, shapes[n], meta['shape'])
        # slots and accumulators live under the scope optimizer.minimize ran in (CenterNet.py:131-156; tests/golden/optimizer_scopes.json)
        assert 'center_detector/backone/conv2d/kernel/Adam_1' in shapes and 'backone/conv2d/kernel/Adam_1' not in shapes
        assert abs(float(reader.get_tensor('center_detector/beta1_power')) - 0.9 ** 12) < 1e-7 and 'beta1_power' not in shapes
        m2 = odtk.CenterNet(dict(cfg, seed=2), prov)
        m2.load_weight(prefix + '-11')
        assert torch.equal(m2.S, m.S) and m2.global_step == 11
        for k in m.pinfo:
            assert torch.equal(m2.get_param(k), m.get_param(k)) and torch.equal(m2.get_param(k, m2.M1), m.get_param(k, m.M1)), k
            assert torch.equal(m2.get_param(k, m2.M2), m.get_param(k, m.M2)), k
        m3 = odtk.CenterNet(dict(cfg, seed=3), prov)
        before = m3.export_params()
        m3.load_pretrained_weight(prefix + '-11')
        after, src = m3.export_params(), m.export_params()
        for k in m.pinfo:
            layer = int(k[1:].split('.')[0])
            assert torch.equal(after[k], src[k] if layer < 50 else before[k]), k


@pytest.mark.parametrize("cls", ["SSD300", "SSD512", "YOLOv3", "RetinaNet", "FCOS", "CenterNet", "RefineDet320", "PFPNetR", "YOLOv2"])
def test_optimizer_slot_scopes_follow_the_reference(cls, tmp_path):
    """Slot variables (`/Momentum`, `/Adam`, `/Adam_1`) and Adam's beta-power accumulators carry the variable scope that is open where the reference
    calls `optimizer.minimize` -- recorded by building the reference's own classes on the shim (tests/golden/optimizer_scopes.json,
    make_golden_optimizer_scopes.py).  Every class must WRITE those names, and must RESTORE the optimizer state from a file that carries them."""
    import json
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import mock_ops
    import odtk
    scope = json.load(open(os.path.join(here, 'golden', 'optimizer_scopes.json')))[cls]
    prefix = scope + '/' if scope else ''
    base = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
            'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'verbose': False, 'compute_dtype': 'f32', 'device': 'cpu',
            'checkpoint_format': 'tf', 'use_graph': False, 'pretraining_weight': ''}
    prov = {'data_shape': [64, 64, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    ctx = contextlib.nullcontext()
    if cls in ('SSD300', 'SSD512'):
        size = 300 if cls == 'SSD300' else 512
        prov = dict(prov, data_shape=[size, size, 3])
        cfg = base
        if cls == 'SSD512':
            from oracle import ssd512_ref as R5
            ctx = R5.tables()
    elif cls == 'YOLOv3':
        from oracle import yolov3_ref as YR
        cfg = dict(base, data_shape=[64, 64, 3], coord_scale=1, noobj_scale=1, obj_scale=5., class_scale=1., num_priors=3, priors=YR.PRIORS_PX)
    elif cls == 'YOLOv2':
        from oracle import yolov2_ref as YR2
        cfg = dict(base, is_pretraining=False, data_shape=[64, 64, 3], coord_scale=1, noobj_scale=1, obj_scale=5., class_scale=1., rescore_confidence=False,
                   priors=YR2.PRIORS)
    elif cls == 'RetinaNet':
        cfg = dict(base, is_bottleneck=True, residual_block_list=[3, 4, 6, 3], init_conv_filters=16, is_pretraining=False, data_shape=[128, 128, 3],
                   gamma=2.0, alpha=0.25)
    elif cls == 'FCOS':
        cfg = dict(base, data_shape=[64, 64, 3])
    elif cls == 'CenterNet':
        cfg = dict(base, input_size=128, score_threshold=0.1, top_k_results_output=10)
        prov = dict(prov, data_shape=[128, 128, 3])
    else:
        cfg = dict(base, input_size=64)
    slot_kinds = ('/Adam', '/Adam_1') if cls == 'CenterNet' else ('/Momentum',)
    with mock_ops.installed(), ctx:
        m = getattr(odtk, cls)(dict(cfg, seed=1), prov)
        state = [m.M1, m.M2] if cls == 'CenterNet' else [m.Mom]
        g = torch.Generator().manual_seed(3)
        for buf in state:
            buf.copy_(torch.rand(buf.shape, generator=g) * 1e-3)
        m.global_step = 5
        out = m.export_tf_variables()
        variables = {k for k in out if not k.endswith(slot_kinds) and not k.endswith(('beta1_power', 'beta2_power'))}
        slots = [k for k in out if k.endswith(slot_kinds)]
        assert slots and all(k.startswith(prefix) and k[len(prefix): k.rindex('/')] in variables for k in slots), (scope, slots[:3])
        if scope:
            assert not any(k[len(prefix):] in out for k in slots), 'a slot is also present without its scope'
        if cls == 'CenterNet':
            assert prefix + 'beta1_power' in out and prefix + 'beta2_power' in out and 'beta1_power' not in out
        path = str(tmp_path / 'ck' / cls)
        m.save_weight('latest', path)
        m2 = getattr(odtk, cls)(dict(cfg, seed=2), prov)
        m2.load_weight(path + '-5')
        out2 = m2.export_tf_variables()               # the optimizer state came back from the scoped names
        assert all(np.array_equal(out[k], out2[k]) for k in slots)


def test_lhrcnn_reference_names_and_saver_round_trip(tmp_path):
    """LHRCNN: the rule-derived names equal those of the graph the reference builds on the shim (tests/golden/lhrcnn_names.json, checked in
    test_models_host_logic_cpu); a `checkpoint_format='tf'` save writes every variable of that graph with its shape (depthwise kernels [kh, kw, C, 1], dense
    kernels [in, units]), the Momentum slots under 'rcnn/' (the scope open where the reference builds its optimizer, LH_RCNN.py:98, :171) and global_step; a
    second model restores all of it bit for bit"""
    import json
    import sys
    import torch
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    import mock_ops
    import odtk
    from odtk.tf_checkpoint import NewCheckpointReader
    cfg = {'data_shape': [320, 416, 3], 'mode': 'train', 'is_pretraining': False, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4,
           'keep_prob': 0.5, 'batch_size': 1, 'rpn_first_step': 60000, 'rcnn_first_step': 100000, 'rpn_second_step': 160000, 'nms_score_threshold': 0.5,
           'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'post_nms_proposal': 500, 'verbose': False, 'device': 'cpu', 'checkpoint_format': 'tf'}
    prov = {'data_shape': [320, 416, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None}
    with mock_ops.installed():
        m = odtk.LHRCNN(dict(cfg, seed=1), prov)
        names = m.reference_variable_map()
        g = torch.Generator().manual_seed(5)
        m.Mom.copy_(torch.randn(m.Mom.shape, generator=g) * 1e-3)
        for k in m.sinfo:
            m.stat(k).copy_(torch.rand(m.stat(k).shape, generator=g) + 0.5)
        m.global_step = 41
        prefix = str(tmp_path / 'ck' / 'lhrcnn')
        m.save_weight('latest', prefix)
        reader = NewCheckpointReader(prefix + '-41')
        shapes = reader.get_variable_to_shape_map()
        tf_vars = json.load(open(os.path.join(here, 'golden', 'lhrcnn_variables.json')))
        assert set(tf_vars) <= set(shapes)
        for n, meta in tf_vars.items():
            assert list(shapes[n]) == meta['shape'], (n, shapes[n], meta['shape'])
        scope = json.load(open(os.path.join(here, 'golden', 'optimizer_scopes.json')))['LHRCNN']       # recorded on the reference's own class
        assert scope == 'rcnn' and m.MOMENTUM_SLOT_SCOPE == scope + '/'
        for ours in ('conv1.w', 'stage3_sconv4.dw', 'rpn_pbbox.gamma', 'state5_conv2_2.w', 'roi_feat_dense.w', 'rcnn_pbbox.b'):
            assert 'rcnn/' + names[ours] + '/Momentum' in shapes, ours
        assert int(reader.get_tensor('global_step')) == 41
        m2 = odtk.LHRCNN(dict(cfg, seed=2), prov)
        assert not torch.equal(m2.P, m.P)
        m2.load_weight(prefix + '-41')
        assert torch.equal(m2.S, m.S) and m2.global_step == 41
        for k in m.pinfo:
            if k.endswith('.b') and k[:-2] in m._sep:
                continue                                                      # the engine's inert bias of a separable layer: not a variable of the graph
            assert torch.equal(m2.get_param(k), m.get_param(k)) and torch.equal(m2.get_param(k, m2.Mom), m.get_param(k, m.Mom)), k
