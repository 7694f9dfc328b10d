"""BASELINE.json's configurations AT THEIR STATED SHAPES on the GPU: every launch of a whole training step of RetinaNet 800x800 batch 16,
YOLOv3 416x416 batch 8 (config 4's per-GPU share), FCOS 512x512 batch 16 and CenterNet 512x512 batch 16 (config 5's per-GPU share) -- and of
SSD300 300x300 batch 32 -- is shadowed in situ (tests/insitu.py): re-executed in plain f32 PyTorch from the engine's own stored inputs of that
launch and compared.  Tile counts, persistent multi-tile walks, split-K decisions, pixel splits of the filter gradients and the 32-bit DMA offsets
all change with the batch and map size, so the toy-shape model tests (tests/test_gpu_*_model.py: 64-160 px, batch 2) do not cover them.
Each class is run on the engine it DEFAULTS to, and on its bf16 engine -- the kernel-level evidence behind every bf16 throughput figure quoted in
BASELINE.md (what bf16 does to the gradients end to end is a property of the arithmetic, characterised separately: tests/test_gpu_engine_bf16.py,
DESIGN.md 5).  The box-side launches are checked in the same pass: loss and d(prediction) against the oracle on the engine's own logits.

Reference shapes: testretinanet.py:22-42, testYOLOv3.py:17-41, testfcos.py:20-32, testcenternet.py."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import insitu          # noqa: E402
import mock_ops        # noqa: E402

CASES = [('retinanet', 'f32'), ('retinanet', 'f32x3'), ('yolov3', 'bf16'), ('fcos', 'bf16'), ('centernet', 'bf16'),        # the engine each class defaults to (steady state)
         ('ssd300', 'bf16'), ('retinanet', 'bf16'), ('yolov3', 'f32'), ('fcos', 'f32'), ('centernet', 'f32')]


# the classes of SURVEY.md 8f.4 at the shapes their throughput is quoted on: SSD512 512x512 b32, RefineDet320 / PFPNetR 320x320 b32, YOLOv2 480x480 b32
CASES += [('ssd512', 'bf16'), ('refinedet', 'bf16'), ('pfpnet', 'bf16'), ('yolov2', 'bf16'), ('refinedet', 'f32'), ('pfpnet', 'f32'), ('yolov2', 'f32')]
# the operand-splitting engine (ODTK_F32X3 descriptors: f32 tensors, three bf16 MFMA products per f32 product where that is faster) of every class that has an f32 engine
CASES += [('refinedet', 'f32x3'), ('pfpnet', 'f32x3'), ('yolov2', 'f32x3'), ('fcos', 'f32x3'), ('centernet', 'f32x3')]
# round 5: the two hand-rolled classes on the descriptor dtype as well (bench.py --dtype f32x3 for the headline class)
CASES += [('ssd300', 'f32x3'), ('yolov3', 'f32x3')]


@pytest.mark.parametrize('name,dtype', CASES, ids=[f'{n}-{d}' for n, d in CASES])
def test_every_launch_in_situ_at_baseline_shape(name, dtype):
    import bench_configs as BC
    torch.set_num_threads(16)
    sh = insitu.Shadow()
    with sh.installed():
        r = BC.make(name, dtype=dtype, use_graph=False)
        m = r['model']
        size0, batch0, _, _ = BC.SHAPES[name]
        assert r['size'] == size0 and r['batch'] == batch0                       # the stated shape, not a reduced one
        if name == 'ssd512':
            from oracle import ssd512_ref as R5
            tables = R5.tables()
            tables.__enter__()                  # the shadowed box-side launches call the oracle, which reads the swapped SSD512 tables
        if name == 'retinanet':
            mock_ops.retina_loss.anchors = tuple(t.cpu() for t in m.anc)
            assert m.num_anchor_boxes == 120087
        m.set_batch(r['images'], r['gt'])
        m.train_step(r['lr'])                     # un-shadowed first step: lazily grown scratch exists, momentum / moving statistics are non-trivial
        sh.recording = True
        loss = m.train_step(r['lr'])
        sh.recording = False
        torch.cuda.synchronize()
        m_desc = list(getattr(m, 'desc', {}).values())
    if name == 'ssd512':
        tables.__exit__(None, None, None)
    assert bool(torch.isfinite(torch.as_tensor(loss)).all())
    rows = sh.check(insitu.default_tol(dtype), verbose=True, label=f'{name} {dtype} {size0}x{size0} batch {batch0}')
    seen = {x['op'] for x in rows}
    assert {'conv2d_fwd', 'conv2d_dgrad', 'conv2d_wgrad'} <= seen and any(o.endswith('_loss') for o in seen)
    if dtype == 'f32x3':                        # the engine is a property of the descriptors: the same launches, most of their FLOPs as split bf16 products
        from odtk import ops
        assert sum(1 for d in m_desc if ops.conv2d_x3_supported(d) & 1) >= 10
    n_conv = len(BC.conv_layers(name, m))
    if n_conv:
        assert sum(1 for x in rows if x['op'] in ('conv2d_fwd', 'conv2d_fwd_pool2x2') and x['out'] in ('y', 'y_pool')) >= n_conv - 1
        assert sum(1 for x in rows if x['op'] == 'conv2d_wgrad' and x['out'] == 'dw') >= n_conv - 1
    del m, r
    torch.cuda.empty_cache()
