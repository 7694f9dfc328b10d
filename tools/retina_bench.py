"""Times the RetinaNet box-side kernels at BASELINE config 3 (800x800, batch 16, 120 087 anchors)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, odtk
from odtk import ops
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as S
dev = torch.device('cuda')
N, ds = int(sys.argv[1]) if len(sys.argv) > 1 else 16, [800, 800, 3]
shapes = S.pyramid_shapes(ds[0], ds[1])
flat = S.retina_priors_flat()
anc = ops.retina_anchors(ds[1], shapes, [9] * 5, flat, dev)
A = anc[0].shape[0]
gt = S.synthetic_gt(N, 800, 3).to(dev)
P = gt.shape[1]
i32 = dict(dtype=torch.int32, device=dev)
ngt = torch.zeros(N, **i32); best = torch.zeros(N, P, **i32); status = torch.zeros(N, A, dtype=torch.uint8, device=dev)
rg = torch.zeros(N, A, **i32); counts = torch.zeros(N, 4, **i32); ws = ops.retina_match_workspace(A, N, P, dev)
pconf = torch.randn(N, A, 21, device=dev); pbox = torch.randn(N, A, 4, device=dev) * 0.5
parts = torch.empty(N, 2, device=dev); dconf = torch.empty_like(pconf); dbox = torch.empty_like(pbox)
def step():
    ops.retina_match(anc[0], anc[1], anc[3], gt, ngt, best, status, rg, counts, ws)
    ops.retina_loss(pconf, pbox, anc[2], anc[3], gt, ngt, best, status, rg, counts, 0.25, 2.0, 1.0 / N, parts, dconf, dbox)
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
reps = 20
tm = tl = 0.0
for _ in range(reps):
    e0.record(); ops.retina_match(anc[0], anc[1], anc[3], gt, ngt, best, status, rg, counts, ws); e1.record()
    ops.retina_loss(pconf, pbox, anc[2], anc[3], gt, ngt, best, status, rg, counts, 0.25, 2.0, 1.0 / N, parts, dconf, dbox); e2.record()
    torch.cuda.synchronize(); tm += e0.elapsed_time(e1); tl += e1.elapsed_time(e2)
byt = N * A * (21 + 4) * 4 * 2 + N * A * (1 + 4)
print(f'A={A} N={N}: match {tm/reps*1e3:.1f} us, focal+smoothL1 fwd+bwd {tl/reps*1e3:.1f} us '
      f'({byt/1e6:.0f} MB algorithmic -> {byt/(tl/reps*1e-3)/1e9:.0f} GB/s), positives {counts[:,0].tolist()[:4]}')
