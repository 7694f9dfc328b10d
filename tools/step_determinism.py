"""Is a class's training step bit-reproducible in deterministic mode (odtk_debug_set key 5)?  Two fresh models per (class, engine), the same seeded weights and
batches, `steps` optimizer steps each; after every step the flat gradient buffers are compared parameter by parameter and the first ones that differ are
named (a non-deterministic kernel shows up as the layers it feeds).

    python tools/step_determinism.py [classes=ssd300,yolov3,fcos,centernet,yolov2,retinanet] [engines=f32,bf16] [steps=3]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch          # noqa: E402

import bench_configs as BC          # noqa: E402


def one(name, engine, steps, batch):
    from odtk import ops
    size = BC.SHAPES[name][0]
    runs = []
    for rep in range(2):
        ops.debug_set(5, 1)
        try:
            r = BC.make(name, batch=batch, size=size, dtype=engine, use_graph=False)          # (an explicit engine: no f32 warm-up twin)
            m = r['model']
            pool = [BC.synthetic_batch(name, batch, size, 100 + i) for i in range(steps)]
            gs, losses = [], []
            for i in range(steps):
                m.set_batch(*pool[i])
                losses.append(float(m.train_step(1e-3)))
                gs.append(m.G.clone())
            torch.cuda.synchronize()
            runs.append((m, gs, losses, m.P.clone()))
        finally:
            ops.debug_set(5, 1)
    (m, g0, l0, p0), (_, g1, l1, p1) = runs
    ok = True
    for s in range(steps):
        if not torch.equal(g0[s], g1[s]):
            ok = False
            bad = []
            for k, (off, shape) in m.pinfo.items():
                n = 1
                for d in shape:
                    n *= int(d)
                a, b = g0[s][off: off + n], g1[s][off: off + n]
                if not torch.equal(a, b):
                    bad.append((k, float((a - b).abs().max()), float(a.abs().max())))
            print(f'  {name} {engine}: step {s}: {len(bad)} of {len(m.pinfo)} gradients differ; first {bad[:6]}; last {bad[-3:]}')
            break
    print(f'DET {name} {engine}: {"bit-identical" if ok and torch.equal(p0, p1) and l0 == l1 else "DIFFERS"} over {steps} steps (losses {l0} | {l1})', flush=True)
    del runs
    torch.cuda.empty_cache()
    return ok


def main():
    names = (sys.argv[1] if len(sys.argv) > 1 else 'ssd300,yolov3,fcos,centernet,yolov2,retinanet').split(',')
    engines = (sys.argv[2] if len(sys.argv) > 2 else 'f32,bf16').split(',')
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for name in names:
        for e in engines:
            try:
                one(name, e, steps, 2 if name == 'retinanet' else 4)
            except Exception as ex:          # noqa: BLE001
                print(f'DET {name} {e}: ERROR {type(ex).__name__}: {ex}', flush=True)


if __name__ == '__main__':
    main()
