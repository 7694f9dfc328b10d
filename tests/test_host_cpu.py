"""CPU: host logic + C-ABI surface (no compute calls without a GPU)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import odtk
    from odtk import _lib
    header = open(os.path.join(ROOT, 'include', 'odtk.h')).read()
    declared = set(re.findall(r'\b(odtk_[a-z0-9_]+)\s*\(', header))
    declared.discard('odtk_filter_to_dgrad')          # mentioned in a comment only
    assert len(declared) >= 25
    lib = _lib.load()                                  # attaches signatures; raises if a symbol is missing
    out = subprocess.check_output(['nm', '-D', _lib.LIB_PATH]).decode()
    exported = set(re.findall(r' T (odtk_[a-z0-9_]+)', out))
    assert declared <= exported, declared - exported
    assert declared <= set(_lib.SIGNATURES), declared - set(_lib.SIGNATURES)
    assert lib.odtk_version() >= 100
    assert isinstance(lib.odtk_last_error(), bytes)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from odtk import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(_lib.OdtkError):
        _lib.load()


def test_no_cpu_fallback_in_model():
    import odtk
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    cfg = {'mode': 'test', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': .5,
           'batch_size': 1, 'nms_score_threshold': .5, 'nms_max_boxes': 20, 'nms_iou_threshold': .5,
           'pretraining_weight': ''}
    with pytest.raises(odtk.OdtkError):
        odtk.SSD300(cfg, None)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, 'object-detection-tensorflow_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'import oracle' not in src and 'from oracle' not in src, f


def test_host_geometry_matches_oracle():
    from odtk import ops
    from odtk.ssd300 import prior_spec, NUM_PRIORS, FEATURE_SIZES
    from oracle import ssd300_ref as R
    assert ops.same_pad(75, 2, 2) == R.same_pad(75, 2, 2) == (38, 0, 1)
    assert ops.same_pad(10, 3, 2) == (5, 0, 1) and ops.same_pad(19, 3, 1, 2) == (19, 2, 2)
    fs, nas, hw = prior_spec()
    assert fs == R.feature_sizes() == FEATURE_SIZES and NUM_PRIORS == 8828
    assert len(hw) == 2 * sum(nas)
    d = ops.conv_desc(32, 10, 10, 128, 128, 256, 256, 3, 2, 1)
    assert (d.Ho, d.Wo, d.pad_t, d.pad_l) == (5, 5, 0, 0)


def test_conv_fastdiv_and_swizzle_properties():
    # mirrors csrc/conv.hip make_fastdiv/fdiv and the LDS swizzles
    def make(d):
        s = 0
        while (1 << s) < d:
            s += 1
        return ((((1 << 32) * ((1 << s) - d)) // d) + 1) & 0xffffffff, s

    def fdiv(x, m, s):
        return (((x * m) >> 32) + x) >> s
    rng = np.random.default_rng(0)
    for d in [1, 2, 3, 5, 9, 19, 38, 75, 150, 300, 361, 1444, 5625, 22500, 90000, 100, 9]:
        m, s = make(d)
        xs = np.concatenate([rng.integers(0, 2 ** 31 - 1, 2000), np.arange(0, 5000), [2 ** 31 - 1, d - 1, d, d + 1]])
        for x in xs.tolist():
            assert fdiv(x, m, s) == x // d, (d, x)

    def swz(row):
        return (((row >> 1) ^ (row >> 5)) & 1) | (((row >> 3) & 1) << 1) | (((row >> 4) & 1) << 2)

    def swz_g(row):
        return ((row >> 1) & 1) | (((row >> 3) & 1) << 1) | (((row >> 4) & 1) << 2)
    groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
              list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
    for f in (swz, swz_g):
        for base in (0, 32, 64, 96):
            for grp in groups:                      # ds_read_b128 lane groups: 16 lanes must hit 16 distinct 16-B slots
                for slot in range(8):
                    pos = {((base + r) & 1) * 8 + (slot ^ f(base + r)) for r in grp}
                    assert len(pos) == 16
    # wgrad DMA piece: 4 rows x 4 chunks read by a half-wave cover all 16 slots of the 256-B bank row
    for ch0 in (0, 4, 8, 12):
        assert len({(ch ^ (r << 2)) for r in range(4) for ch in range(ch0, ch0 + 4)}) == 16


def test_augmentor_plan_matches_oracle_plan():
    """odtk.augment.Augmentor.plan (host logic: sizes, ratios, draw order) against oracle/augment_ref.plan, no GPU"""
    import numpy as np
    from odtk import augment as A
    from oracle import augment_ref as AR
    rng = np.random.default_rng(3)
    cfgs = [dict(output_shape=[300, 300], crop_method='random', flip_prob=[0., 0.5], fill_mode='BILINEAR', keep_aspect_ratios=False,
                 color_jitter_prob=0.5, rotate=[0.5, -5., -5.]),
            dict(output_shape=[64, 96], zoom_size=[80, 120], crop_method='random', flip_prob=[0.5, 0.5], fill_mode='BILINEAR',
                 keep_aspect_ratios=True, constant_values=1., color_jitter_prob=0.7, rotate=[0.6, -5., 5.]),
            dict(output_shape=[48, 48], zoom_size=[56, 60], crop_method='center', fill_mode='BILINEAR'),
            dict(output_shape=[72, 80], fill_mode='CONSTANT', flip_prob=[0.5, 0.5]),
            dict(output_shape=[64, 96], zoom_size=[80, 120], crop_method='random', fill_mode='NEAREST_NEIGHBOR', keep_aspect_ratios=True,
                 constant_values=1.),
            dict(output_shape=[48, 48], zoom_size=[56, 60], crop_method='center', fill_mode='BICUBIC', color_jitter_prob=0.3)]
    for cfg in cfgs:
        aug = A.Augmentor('channels_last', **cfg)
        assert aug.plan(33, 44, A._Draws([0] * 14))['resize'] == {'CONSTANT': 0, 'BILINEAR': 1, 'NEAREST_NEIGHBOR': 2, 'BICUBIC': 3}[cfg['fill_mode']]
        for _ in range(50):
            h, w = int(rng.integers(20, 600)), int(rng.integers(20, 600))
            draws = [int(rng.integers(0, 8)), int(rng.integers(0, 8))] if cfg.get('zoom_size') and cfg['crop_method'] == 'random' else []
            draws += [float(rng.uniform()) for _ in range(12)]
            d = A._Draws(list(draws))
            mine = aug.plan(h, w, d)
            ref = AR.plan([h, w, 3], cfg['output_shape'], cfg.get('zoom_size'), cfg.get('crop_method'), cfg.get('flip_prob'),
                          cfg['fill_mode'], cfg.get('keep_aspect_ratios', False), cfg.get('color_jitter_prob'), cfg.get('rotate'), draws)
            assert d.q == ref['unused_draws']
            for k in ('resize_h', 'resize_w', 'crop_h', 'crop_w', 'ratio_y', 'ratio_x'):
                assert mine[k] == ref[k], (k, mine[k], ref[k])
            assert bool(mine['flip_td']) == ref['flip_td'] and bool(mine['flip_lr']) == ref['flip_lr']
            for k in ('brightness', 'contrast', 'hue'):
                assert (ref[k] is not None) == bool(mine['has_' + k]) and (ref[k] is None or ref[k] == mine[k])
            assert (ref['angle'] is not None) == bool(mine['has_rotate']) and (ref['angle'] is None or ref['angle'] == mine['angle'])


def test_augmentor_resize_arithmetic_matches_oracle(tmp_path):
    """csrc/augment_resize.h (what aug_geometry_kernel executes for NEAREST_NEIGHBOR / BICUBIC: scale, source indices, the four
    bicubic weights and their clamped taps) compiled for the host with g++ and compared with oracle/augment_ref.py: indices equal,
    weights bit-equal, for up- and down-scaling, sizes of 1 and the sizes of the driver scripts"""
    import ctypes as C
    import os
    import subprocess
    import numpy as np
    from oracle import augment_ref as AR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = str(tmp_path / 'libaugresize.so')
    subprocess.check_call(['g++', '-O2', '-std=c++17', '-ffp-contract=off', '-shared', '-fPIC', '-I',
                           os.path.join(root, 'object-detection-tensorflow_amd', 'csrc'), os.path.join(root, 'tests', 'augment_resize_host.cpp'), '-o', so])
    lib = C.CDLL(so)
    lib.odtk_test_resize_scale.restype = C.c_float
    rng = np.random.default_rng(11)
    sizes = [(3, 5), (2, 4), (6, 8), (375, 300), (500, 300), (333, 416), (37, 1), (1, 9), (300, 300), (97, 512), (1024, 77)]
    sizes += [(int(rng.integers(1, 700)), int(rng.integers(1, 700))) for _ in range(60)]
    for n_in, n_out in sizes:
        if n_out > 1:
            assert np.float32(lib.odtk_test_resize_scale(n_in, n_out)) == AR._resize_scale(n_in, n_out), (n_in, n_out)
        idx = np.zeros(n_out, np.int32)
        lib.odtk_test_nearest_indices(n_in, n_out, idx.ctypes.data_as(C.c_void_p))
        img = np.arange(n_in, dtype=np.float32).reshape(n_in, 1, 1)
        import torch
        want = AR.resize_nearest_align(torch.from_numpy(img), n_out, 1).reshape(-1).numpy().astype(np.int32)
        assert np.array_equal(idx, want), (n_in, n_out)
        w = np.zeros((n_out, 4), np.float32)
        ti = np.zeros((n_out, 4), np.int32)
        lib.odtk_test_bicubic_taps(n_in, n_out, w.ctypes.data_as(C.c_void_p), ti.ctypes.data_as(C.c_void_p))
        ww, wi = AR._bicubic_taps(n_in, n_out)
        assert np.array_equal(ti, wi.astype(np.int32)), (n_in, n_out)
        assert np.array_equal(w.view(np.uint32), ww.view(np.uint32)), (n_in, n_out, float(np.abs(w - ww).max()))


def test_retinanet_layer_specs_and_priors_match_oracle_and_reference_graph():
    """odtk.retinanet.layer_specs / level_priors (host logic) against the oracle's and against the variables of the reference's own
    class (tests/golden/retinanet_variables.json: 122 kernels with these shapes, in this order)"""
    import json
    import os
    from odtk import retinanet as R
    from oracle import retinanet_net_ref as NR
    from oracle import retinanet_ref as RR
    specs = R.layer_specs([3, 4, 6, 3], 16, 21, 9)
    assert specs == NR.layer_specs()
    for s in RR.ANCHOR_SIZES:
        assert [tuple(map(float, v)) for v in R.level_priors(s)] == [tuple(map(float, v)) for v in RR.level_priors(s)]
    want = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'retinanet_variables.json')))
    # creation order = the order of reference_variable_map (default layer names are numbered per enclosing variable scope)
    vm = R.reference_variable_map()
    assert set(vm) | {'global_step'} == set(want)
    kernels = [k for k in vm if k.endswith('/kernel')]
    gammas = [k for k in vm if k.endswith('/gamma')]
    assert len(kernels) == len(gammas) == len(specs) == 122
    for i, ((name, cin, cout, k, _, bnc, _), kn, gn) in enumerate(zip(specs, kernels, gammas)):
        assert vm[kn] == f'l{i}.w' and vm[gn] == f'l{i}.gamma'
        assert want[kn]['shape'] == [k, k, cin, cout], (name, kn)
        assert want[gn]['shape'] == [bnc], (name, gn)
    assert 'feature_extractor/conv2d_7/kernel' in want and 'regressor/conv2d_49/kernel' in want and 'regressor/conv2d_50/kernel' not in want


def test_f32x3_descriptor_policy_is_host_logic():
    """`odtk_conv2d_x3_supported` (include/odtk.h): which passes of an ODTK_F32X3 descriptor run as split bf16 products is decided on the host -- no GPU needed.
    bit 0 = forward, bit 1 = filter gradient, bit 2 = input gradient."""
    import odtk  # noqa: F401
    from odtk import ops

    def bits(C_, K, k, stride=1, H=50, dtype=ops.F32X3, N=2):
        return ops.conv2d_x3_supported(ops.conv_desc(N, H, H, C_, C_, K, ops.pad_to(K, 4), k, stride, 1, dtype, dtype))
    assert bits(256, 256, 3) == 7                      # the pyramid / head layers of RetinaNet.py:594-643: forward, filter gradient, input gradient
    assert bits(256, 189, 3) == 7 and bits(256, 36, 3) == 7
    assert bits(256, 256, 3, dtype=ops.F32) == 0       # an exact-f32 descriptor never splits
    assert bits(8, 28, 1) == 0 and bits(56, 14, 1) == 0 and bits(28, 28, 3) == 0        # C K R S < 20 000: the narrow backbone layers stay exact ...
    assert bits(28, 56, 3, stride=2) == 4 and bits(16, 14, 3, stride=2) == 4            # ... but for the input gradient of a 3x3 stride-2 layer
    assert bits(28, 256, 3) == 5                       # 28 channels: forward / input gradient split, the filter gradient (rows of C) exact
    assert bits(256, 256, 3, stride=2) == 7 and bits(256, 256, 3, stride=3) == 0
    assert bits(2048, 21, 1, H=1, N=8192) == 7         # LH_RCNN's dense head (rows as 1 x 1 images)
    # the split operands are addressed with 32-bit byte offsets: a map whose [hi | lo] copy passes 2 GiB stays exact
    assert bits(256, 256, 3, H=1100, N=8) == 0


def test_f32x3_descriptors_are_refused_by_the_fused_entry_points():
    """ODTK_F32X3 is a property of the three plain convolution passes: the fused conv + pool / sign-bit entry points say so (argument checks run before any launch: no GPU needed)"""
    import odtk  # noqa: F401
    from odtk import _lib, ops
    d = ops.conv_desc(2, 64, 64, 64, 64, 64, 64, 3, 1, 1, ops.F32X3, ops.F32X3)
    assert ops.conv2d_fwd_pool2x2_fused(d) is False or ops.conv2d_fwd_pool2x2_fused(d) == 0
    assert not ops.conv2d_relu_bits_supported(d, d, 64)
    lib = _lib.load()
    import ctypes as C
    rc = lib.odtk_conv2d_fwd_bits(C.byref(d), None, None, None, None, 1, None, None)
    assert rc != 0 and b'ODTK_F32X3' in lib.odtk_last_error()
    # a descriptor that mixes the engine with another storage type is refused as well
    bad = ops.conv_desc(2, 64, 64, 64, 64, 64, 64, 3, 1, 1, ops.F32X3, ops.F32)
    assert lib.odtk_conv2d_fwd(C.byref(bad), None, None, None, None, 0, None) != 0 and b'ODTK_F32X3' in lib.odtk_last_error()


def test_comm_entry_points_check_their_arguments_before_touching_rccl():
    """include/odtk.h: odtk_comm_* (the C-ABI's collective).  Argument checks run before RCCL is bound or a device is touched: no GPU needed"""
    import ctypes as C
    import odtk  # noqa: F401
    from odtk import _lib
    lib = _lib.load()
    assert lib.odtk_comm_unique_id(None) == 1 and b'null' in lib.odtk_last_error()
    ident = C.create_string_buffer(128)
    comm = C.c_void_p()
    assert lib.odtk_comm_init(ident, 2, 2, C.byref(comm)) == 1 and b'rank 2 of 2' in lib.odtk_last_error()
    assert lib.odtk_comm_init(None, 0, 1, C.byref(comm)) == 1
    assert lib.odtk_comm_allreduce(None, None, None, 0, _lib.F32, None) == 1 and b'null communicator' in lib.odtk_last_error()
    assert lib.odtk_comm_broadcast(None, None, 0, _lib.F32, 0, None) == 1
    assert lib.odtk_comm_info(None, None, None) == 1
    assert lib.odtk_comm_destroy(None) == 0            # destroying nothing is not an error
    from odtk.dist import COMM_ID_BYTES
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'odtk.h')).read()
    assert f'#define ODTK_COMM_ID_BYTES {COMM_ID_BYTES}' in hdr


def test_bn_workspace_holds_the_partials_of_every_plan():
    """odtk_bn_workspace_bytes (host logic): two sums x the row splits a launch may use x the padded channels, + the finalized scale / offset rows.  From 256
    channels on that is 256 splits; narrow layers get the same 2 x 256 x 256 floats, i.e. up to 1 024 splits for one column group (round 5)."""
    import odtk  # noqa: F401
    from odtk import _lib
    lib = _lib.load()
    for C in (3, 8, 16, 32, 64, 100, 128, 150, 255, 256, 257, 512, 1000, 1024, 2048):
        cpad = (C + 63) // 64 * 64
        b = lib.odtk_bn_workspace_bytes(1 << 20, C)
        assert b == lib.odtk_bn_workspace_bytes(7, C)                     # independent of the row count
        assert b >= (2 * 256 + 2) * cpad * 4                             # what every release of the library has needed
        rows = min(1024, 256 * max(1, 256 // cpad) * cpad // C)          # row splits the statistics kernels may use for this width
        assert b >= (2 * rows * C + 2 * cpad) * 4, (C, b, rows)
        if cpad in (64, 128):                                             # (192 = 3 x 64 does not divide 256: it keeps 256 rows)
            assert b >= 2 * 256 * 256 * 4
