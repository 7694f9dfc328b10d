import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, odtk
from odtk import ops
dev = torch.device('cuda')
n, thr, max_out = 3000, 0.3, 100
g = torch.Generator().manual_seed(n + max_out)
B = 3
yx = torch.rand(B, n, 2, generator=g) * 300
hw = torch.rand(B, n, 2, generator=g) * 80 + 5
boxes = torch.cat([yx - hw / 2, yx + hw / 2], -1).contiguous()
boxes[0, : n // 3] = boxes[0, : n // 3][:, [2, 3, 0, 1]]
scores = torch.stack([(torch.randperm(n, generator=g).float() + 1) / n for _ in range(B)])
valid = (torch.rand(B, n, generator=g) > 0.2).to(torch.uint8) * 2
res = {}
for eng in (1, 0):
    ops.debug_set(3, eng)
    out_idx = torch.full((B, max_out), -1, dtype=torch.int32, device=dev)
    out_cnt = torch.zeros(B, dtype=torch.int32, device=dev)
    mo = torch.tensor([max_out, max(max_out // 2, 1), 0], dtype=torch.int32, device=dev)
    ops.nms_batched(boxes.to(dev), n * 4, scores.to(dev), n, 1, valid.to(dev), n, 1, 2, n, B, mo, 1, 0, thr, out_idx, max_out, out_cnt)
    torch.cuda.synchronize()
    res[eng] = (out_idx.cpu(), out_cnt.cpu())
    print('engine', eng, 'cnt', out_cnt.cpu().tolist())
a, b = res[1], res[0]
for i in range(B):
    ca, cb = int(a[1][i]), int(b[1][i])
    la, lb = a[0][i, :ca].tolist(), b[0][i, :cb].tolist()
    first = next((k for k in range(min(ca, cb)) if la[k] != lb[k]), None)
    print('problem', i, 'single', ca, 'split', cb, 'first diff at', first)
    if first is not None:
        print(' single', la[max(0, first - 2): first + 4]); print(' split ', lb[max(0, first - 2): first + 4])
# analyse problem 0
import numpy as np
i = 0
m = valid[i] == 2
sc = scores[i].clone(); sc[~m] = -1
order = torch.argsort(sc, descending=True, stable=True)[: int(m.sum())].tolist()
pos = {idx: k for k, idx in enumerate(order)}
la = a[0][i, :int(a[1][i])].tolist(); lb = b[0][i, :int(b[1][i])].tolist()
first = next(k for k in range(len(la)) if la[k] != lb[k])
def iou(p, q):
    bp = boxes[i, p].tolist(); bq = boxes[i, q].tolist()
    y0, y1 = min(bp[0], bp[2]), max(bp[0], bp[2]); x0, x1 = min(bp[1], bp[3]), max(bp[1], bp[3])
    v0, v1 = min(bq[0], bq[2]), max(bq[0], bq[2]); u0, u1 = min(bq[1], bq[3]), max(bq[1], bq[3])
    ih = max(min(y1, v1) - max(y0, v0), 0); iw = max(min(x1, u1) - max(x0, u0), 0)
    inter = ih * iw
    return inter / ((y1 - y0) * (x1 - x0) + (v1 - v0) * (u1 - u0) - inter)
print('nvalid', len(order), 'first diff pick', first, 'single picks', la[first], 'at sorted pos', pos[la[first]], '; split picks', lb[first], 'at pos', pos[lb[first]])
bad = lb[first]
for k in range(first):
    v = iou(la[k], bad)
    if v > thr: print('  split pick', bad, '(pos', pos[bad], ') should be suppressed by pick', k, 'idx', la[k], 'pos', pos[la[k]], 'iou', v)
print('positions of single picks around', [pos[x] for x in la[first-3:first+3]])
print('positions of split picks around', [pos[x] for x in lb[first-3:first+3]])
