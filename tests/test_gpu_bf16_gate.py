"""The gate behind `compute_dtype` defaulting to the bf16 engine for SSD300, FCOS, CenterNet and YOLOv2 -- and NOT for YOLOv3 and RetinaNet (round-2 review, item 6;
round-5 review, items 1-2; object-detection-tensorflow_amd/warmup.py).  Since round 6 the whole file is deterministic: see _gate().

At random initialisation the bf16 engine's filter gradients of these identity-free conv + norm stacks keep the norm and lose the direction towards the input
(cosine against the f32 engine 0.3-0.5 on the input-side third of the layers; SSD300, with its un-normalised VGG trunk, 0.86).  The class is trained 600 optimizer
steps on its f32 engine (synthetic VOC-shaped batches at the BASELINE resolution) and the comparison is repeated FROM THOSE WEIGHTS after 300 and after 600 steps on
16 held-out images: the direction must be back and stay back (every layer > 0.8, input-side third > 0.88 at both) for the bf16 engine to be the training default --
which is then preceded by exactly such a warm-up (300 steps on an f32x3 twin) when a run starts from random initialisation.  YOLOv3 sits AT the bar (0.895 after
300 steps, 0.874 after 600) and RetinaNet far below it (0.54): they train on 'f32x3'; so do RefineDet320 and PFPNetR (profiles/r03n_bf16_after_training_8f4.md).
Measured numbers: profiles/r06_bf16_gate_table.md (round 3's, taken with float-atomic filter gradients: profiles/r03i_bf16_after_training.md)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))


BAR_MIN, BAR_THIRD = 0.8, 0.88
GATE_STEPS, GATE_PROBES = (300, 600), 4


def _gate(name):
    """The gate, as of round 6 (round-5 review, items 1 and 2).  DETERMINISTIC: the 600 f32 steps and every comparison step run with the library's fixed-order
    filter-gradient reduction (the default since round 6; tools/bf16_after_training.py switches it on regardless), seeded weights and batches -- the trained state
    is bit-identical from run to run and from box to box (the sha256 of the trained parameters is printed; profiles/r06_bf16_gate_table.md lists it from three
    boxes), so every number below is THE number of this code, not a draw.  The compared gradient is the sum over four held-out batches (16 images).  Admitted =
    every layer > 0.8 and the input-side third > 0.88 after 300 AND after 600 f32 steps: the direction must be back and must STAY back as training goes on."""
    import bf16_after_training as T
    r = T.run(name, steps=GATE_STEPS[-1], batch=4, lr=1e-3, verbose=True, checkpoints=GATE_STEPS, probes=GATE_PROBES)
    ok = [r['table'][s][0] > BAR_MIN and r['table'][s][1] > BAR_THIRD for s in GATE_STEPS]
    print(f'GATE {name}: init {r["init"][0]:.4f} / {r["init"][1]:.4f}; ' + '; '.join(f'{s}: {r["table"][s][0]:.4f} / {r["table"][s][1]:.4f} state {r["table"][s][2]}'
                                                                                     for s in GATE_STEPS) + f'; admitted: {all(ok)}')
    return r, all(ok)


@pytest.mark.parametrize('name', ['ssd300', 'fcos', 'centernet', 'yolov2'])
def test_bf16_gradients_recover_after_f32_training(name):
    """The classes whose training default is the bf16 engine behind a 300-step f32x3 warm-up.  Measured (r06, identical on three boxes), minimum / input-side third
    at initialisation -> after 300 -> after 600 steps: SSD300 0.789 / 0.862 -> 0.833 / 0.915 -> 0.937 / 0.970; FCOS 0.144 / 0.356 -> 0.890 / 0.924 -> 0.905 / 0.940;
    CenterNet 0.387 / 0.413 -> 0.874 / 0.891 -> 0.891 / 0.912; YOLOv2 0.508 / 0.523 -> 0.920 / 0.929 -> 0.922 / 0.929."""
    import odtk
    r, admitted = _gate(name)
    assert r['init'][1] < BAR_THIRD + 0.0 or r['init'][0] < BAR_MIN, r['init']          # the problem exists at initialisation: hence the warm-up ...
    assert admitted, r['table']                                                         # ... and is gone, and stays gone, after it
    assert abs(r['loss_bf16'] - r['loss_f32']) <= 2e-2 * abs(r['loss_f32'])
    # hence the class default: bf16 engine, f32x3 warm-up of 300 steps when no engine is named and no weights are loaded
    import bench_configs as BC
    cfg, size, batch, _ = BC.config_of(name, batch=1, size=300 if name == 'ssd300' else 128)
    cfg.pop('compute_dtype')
    cls = {'ssd300': odtk.SSD300, 'fcos': odtk.FCOS, 'centernet': odtk.CenterNet, 'yolov2': odtk.YOLOv2}[name]
    m = cls(cfg, {'data_shape': [size, size, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    assert m.DT == odtk.ops.BF16 and m.f32_warmup_steps == 300
    m.load_oracle_params(m.export_params())                                             # loading weights cancels the warm-up
    assert m.f32_warmup_steps == 0


def test_yolov3_sits_at_the_bar_and_trains_on_f32x3_by_default():
    """YOLOv3 through the same gate (round-5 review, item 2): 0.468 / 0.484 at initialisation, 0.854 / 0.895 after 300 steps, 0.853 / 0.874 after 600 -- AT the bar and
    not moving away from it (300 steps clear it by 0.015, 600 steps miss it by 0.006; every other class keeps rising).  Not admitted: the class default for
    training is 'f32x3'; 'bf16' is an explicit choice.  If this test ever fails because the class clears the bar with room, YOLOv3 can move to bf16 + warm-up."""
    import odtk
    r, admitted = _gate('yolov3')
    assert r['init'][1] < 0.65, r['init']
    assert min(r['table'][s][1] for s in GATE_STEPS) < BAR_THIRD + 0.02, r['table']     # does not clear the bar with room at both checkpoints
    assert min(r['table'][s][0] for s in GATE_STEPS) > 0.75 and min(r['table'][s][1] for s in GATE_STEPS) > 0.8, r['table']     # ... it is AT the bar, not far below
    import bench_configs as BC
    cfg, size, batch, _ = BC.config_of('yolov3', batch=1, size=128)
    cfg.pop('compute_dtype')
    m = odtk.YOLOv3(cfg, {'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    assert m.DT == odtk.ops.F32 and m.CDT == odtk.ops.F32X3 and m.f32_warmup_steps == 0
    mb = odtk.YOLOv3(dict(cfg, compute_dtype='bf16'), {'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    assert mb.DT == odtk.ops.BF16 and mb.f32_warmup_steps == 0                         # explicit engine: taken literally


def test_retinanet_does_not_pass_the_gate_and_keeps_f32():
    import bf16_after_training as T
    import odtk
    r = T.run('retinanet', steps=300, batch=2, lr=1e-3, verbose=True)
    assert r['after'][1] > r['init'][1] + 0.15                  # it does recover (deterministic since round 6: 0.005 -> 0.601, one held-out batch) ...
    assert r['after'][1] < 0.9                                  # ... but not to the bar after 300 steps; if this ever fails, RetinaNet can move to bf16 + warm-up too
    import bench_configs as BC
    cfg, size, batch, _ = BC.config_of('retinanet', batch=1, size=128)
    cfg.pop('compute_dtype')
    m = odtk.RetinaNet(cfg, {'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    assert m.DT == odtk.ops.F32 and m.x3          # f32 tensors; since round 4 its large convolutions run as three bf16 products (next test)
    mt = odtk.RetinaNet(dict(cfg, mode='test'), None)
    assert mt.DT == odtk.ops.F32 and not mt.x3     # inference: the exact f32 kernels


def test_retinanet_operand_splitting_engine_passes_the_gate_without_a_warm_up():
    """'f32x3' (three bf16 MFMA products per f32 product, include/odtk.h) through the SAME gate: at RANDOM INITIALISATION and the BASELINE resolution -- where the
    bf16 engine's input-side gradients have cosine ~0.0 against the f32 engine's -- every filter gradient already clears the bar the bf16 engines need 300 f32 steps
    for (every layer > 0.8, input-side third > 0.9; measured 0.972 / 0.979), and after those steps it is indistinguishable from the f32 engine."""
    import bf16_after_training as T
    r = T.run('retinanet', steps=300, batch=2, lr=1e-3, verbose=True, engine='f32x3')
    assert r['init'][0] > 0.9 and r['init'][1] > 0.95, r['init']
    assert r['after'][0] > 0.99 and r['after'][1] > 0.999, r['after']
    assert abs(r['loss_bf16'] - r['loss_f32']) <= 1e-3 * abs(r['loss_f32'])
