"""Is the bf16 engine usable for the batch-norm / group-norm classes AWAY from random initialisation?  (round-2 review, item 6)

At random initialisation the bf16 engine's gradients of RetinaNet / FCOS / CenterNet lose their direction towards the input (the identity-free
conv + norm stacks amplify the 2^-9 rounding of every stored activation), which is why these classes default to the f32 engine.  This tool trains the class on
its f32 engine for `steps` optimizer steps (synthetic VOC-shaped batches, the BASELINE resolution, a reduced batch), then -- from THOSE weights and moving
statistics -- runs one step on a held-out batch on both engines and compares every filter gradient: cosine and norm ratio, bf16 against f32.  The same
comparison from the initial weights is printed next to it.

    python tools/bf16_after_training.py retinanet|fcos|centernet|yolov3 [steps=300] [batch=4] [lr=...]
"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch          # noqa: E402

import bench_configs as BC          # noqa: E402


def grads_of(name, params, stats, batch, size, dtype, probe):
    """`probe`: one held-out batch or a LIST of them -- the filter gradients are summed over the list (the gradient of the mean loss over len(list) x batch held-out
    images, evaluated `batch` images at a time; moving statistics restored in front of every one); -> (mean loss, {name: gradient})"""
    probes = probe if isinstance(probe, list) else [probe]
    sim = dtype == 'f32x1sim'          # EXPERIMENT: the x3 engine with the low halves of both operands zeroed (odtk_debug_set key 6 bits 17 + 3) = f32 tensors, ONE bf16 product
    if sim:
        from odtk import ops
        ops.debug_set(6, (1 << 17) | 8)
    try:
        return _grads_of(name, params, stats, batch, size, 'f32x3' if sim else dtype, probes)
    finally:
        if sim:
            ops.debug_set(6, 0)


def _grads_of(name, params, stats, batch, size, dtype, probes):
    r = BC.make(name, batch=batch, size=size, dtype=dtype, use_graph=False)
    m = r['model']
    m.load_oracle_params(params)
    loss, g = 0.0, None
    for pb in probes:
        if stats is not None and hasattr(m, 'S'):
            m.S.copy_(stats.to(m.S.device))
        m.set_batch(*pb)
        loss += float(m.train_step(0.0)) / len(probes)
        torch.cuda.synchronize()
        gi = {k: m.get_param(k, m.G).double().cpu() for k in m.pinfo if k.endswith('.w')}
        g = gi if g is None else {k: g[k] + gi[k] for k in g}
    del m
    torch.cuda.empty_cache()
    return loss, g


def compare(gf, gb):
    rows = []
    for k, a in gf.items():
        b = gb[k]
        na, nb = float(a.norm()), float(b.norm())
        if na == 0.0:
            continue
        rows.append((k, float((a * b).sum() / (na * nb + 1e-300)), nb / na))
    return rows


def summarize(tag, rows):
    cos = [r[1] for r in rows]
    third = max(1, len(rows) // 3)
    first, last = rows[:third], rows[-third:]
    print(f'{tag}: {len(rows)} filter gradients; cosine against the f32 engine: min {min(cos):.3f} ({min(rows, key=lambda r: r[1])[0]}), median {statistics.median(cos):.3f}; '
          f'first third of the layers (towards the input) median {statistics.median(r[1] for r in first):.3f}, last third median '
          f'{statistics.median(r[1] for r in last):.3f}; norm ratio median {statistics.median(r[2] for r in rows):.3f}')
    return min(cos), statistics.median(r[1] for r in first)


def state_hash(params, stats):
    """sha256 over the exported parameters (and moving statistics): equal hashes <=> bit-identical training runs"""
    import hashlib
    h = hashlib.sha256()
    for k in params:
        h.update(k.encode())
        h.update(params[k].detach().cpu().contiguous().numpy().tobytes())
    if stats is not None:
        h.update(stats.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:16]


def run(name, steps=300, batch=4, lr=1e-3, verbose=True, engine='bf16', deterministic=True, checkpoints=None, probes=1):
    """-> dict(init=(min cosine, input-side-third median), after=(...), losses=[...], hash=..., table={steps: (min, third, hash)})

    `deterministic` (default since round 6): the library's fixed-order filter-gradient reduction (odtk_debug_set key 5) for the f32 training AND both
    comparison steps, so that the trained weights -- and with them every number returned -- are bit-identical from run to run and from box to box
    (`hash` says so: sha256 of the trained parameters + moving statistics).  `checkpoints`: the comparison is repeated at each of these step counts
    (the last one is `steps`).  `probes`: held-out batches the compared gradients are summed over (grads_of)."""
    from odtk import ops
    ops.debug_set(5, 1 if deterministic else 0)
    try:
        return _run(name, steps, batch, lr, verbose, engine, sorted(set(list(checkpoints or []) + [steps])), probes)
    finally:
        ops.debug_set(5, 1)                                # the library's default since round 6


def _run(name, steps, batch, lr, verbose, engine, checkpoints, probes=1):
    size = BC.SHAPES[name][0]
    r = BC.make(name, batch=batch, size=size, dtype='f32', use_graph=False)
    m = r['model']
    probe = BC.synthetic_batch(name, batch, size, 4242) if probes == 1 else [BC.synthetic_batch(name, batch, size, 4242 + 17 * i) for i in range(probes)]
    p0 = m.export_params()
    s0 = m.S.clone() if hasattr(m, 'S') else None
    lf, gf = grads_of(name, p0, s0, batch, size, 'f32', probe)
    lb, gb = grads_of(name, p0, s0, batch, size, engine, probe)
    if verbose:
        print(f'{name} {size}x{size} batch {batch}: at initialisation loss f32 {lf:.4f} / {engine} {lb:.4f}   (state {state_hash(p0, s0)})')
    init = summarize('  initial weights', compare(gf, gb))
    pool = [BC.synthetic_batch(name, batch, size, 100 + i) for i in range(8)]
    losses, table, after = [], {}, None
    for i in range(steps):
        m.set_batch(*pool[i % len(pool)])
        loss = m.train_step(lr)
        if i % max(1, steps // 6) == 0 or i == steps - 1:
            losses.append((i, round(float(loss), 4)))
        if i + 1 in checkpoints:
            torch.cuda.synchronize()
            p1 = m.export_params()
            s1 = m.S.clone() if hasattr(m, 'S') else None
            h = state_hash(p1, s1)
            lf, gf = grads_of(name, p1, s1, batch, size, 'f32', probe)
            lb, gb = grads_of(name, p1, s1, batch, size, engine, probe)
            if verbose:
                print(f'  after {i + 1} steps: held-out loss f32 {lf:.4f} / {engine} {lb:.4f}   (state {h})')
            after = summarize(f'  after {i + 1} steps', compare(gf, gb))
            table[i + 1] = (after[0], after[1], h)
    if verbose:
        print(f'  f32 training, lr {lr}: loss {losses}')
    del m
    torch.cuda.empty_cache()
    print(f'RESULT {name}: min cosine {init[0]:.3f} -> {after[0]:.3f}; input-side third {init[1]:.3f} -> {after[1]:.3f}; state {table[steps][2]}')
    return dict(init=init, after=after, losses=losses, loss_f32=lf, loss_bf16=lb, hash=table[steps][2], table=table)


def compare_engines(name, a='f32', b='f32x3', batch=2, size=None, verbose=True):
    """One step from the SAME (initial) weights on engines a and b: -> (min cosine, input-side-third median cosine, |loss_b - loss_a| / loss_a) over every filter
    gradient.  Initialisation is the worst case for these stacks (see the module docstring); an engine that keeps the direction there keeps it everywhere."""
    size = size or BC.SHAPES[name][0]
    r = BC.make(name, batch=batch, size=size, dtype='f32', use_graph=False)
    m = r['model']
    probe = BC.synthetic_batch(name, batch, size, 4242)
    p0 = m.export_params()
    s0 = m.S.clone() if hasattr(m, 'S') else None
    del m
    torch.cuda.empty_cache()
    la, ga = grads_of(name, p0, s0, batch, size, a, probe)
    lb, gb = grads_of(name, p0, s0, batch, size, b, probe)
    rows = compare(ga, gb)
    if verbose:
        print(f'{name} {size}x{size} batch {batch}: loss {a} {la:.5f} / {b} {lb:.5f}')
    mn, third = summarize(f'  {b} against {a}, initial weights', rows)
    return mn, third, abs(lb - la) / abs(la)


def main():
    name = sys.argv[1]
    if len(sys.argv) > 2 and sys.argv[2] == 'x3':                       # the operand-splitting engine through the same gate
        run(name, 300, int(sys.argv[3]) if len(sys.argv) > 3 else 2, 1e-3, engine='f32x3')
        return
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    lr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
    run(name, steps, batch, lr)


if __name__ == '__main__':
    main()
