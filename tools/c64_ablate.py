"""Where does conv3x3_c64k64_kernel (conv1_2 forward / input gradient at batch 32) spend its time?  Ablation runs (results are garbage with a bit set):
bit 0 = no patch DMA after the first tile, bit 10 = no fragment reads / MFMAs, bit 6 = no epilogue.  usage: python tools/c64_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from odtk import ops
dev = torch.device('cuda')
N, H = 32, 300
d = ops.conv_desc(N, H, H, 64, 64, 64, 64, 3, 1, 1)
x = torch.randn(N * H * H, 64, device=dev).to(torch.bfloat16)
w = (torch.randn(64, 3, 3, 64, device=dev) * 0.05)
wc = w.to(torch.bfloat16).contiguous()
wt = torch.empty(64 * 9 * 64, dtype=torch.bfloat16, device=dev)
ops.filter_prepare(w, 64, 3, 3, 64, 64, ops.BF16, None, wt)
b = torch.zeros(64, device=dev)
y = torch.zeros(N * H * H, 64, dtype=torch.bfloat16, device=dev)
yp = torch.zeros(N * 150 * 150, 64, dtype=torch.bfloat16, device=dev)
idx = torch.zeros(N * 150 * 150 * 8, dtype=torch.int16, device=dev)
dy = torch.randn(N * H * H, 64, device=dev).to(torch.bfloat16)
dx = torch.empty_like(x)


def t(fn, reps=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


cases = {'fwd (full store)': lambda: ops.conv2d_fwd(d, x, wc, b, y, True),
         'fwd + pool, keep y': lambda: ops.conv2d_fwd_pool2x2(d, x, wc, b, y, True, yp, idx),
         'fwd + pool only': lambda: ops.conv2d_fwd_pool2x2(d, x, wc, b, None, True, yp, idx),
         'dgrad (mask)': lambda: ops.conv2d_dgrad(d, dy, 64, wt, x, dx, False),
         'dgrad (no mask)': lambda: ops.conv2d_dgrad(d, dy, 64, wt, None, dx, False)}
head = os.environ.get('ODTK_LIB', '').endswith('_head.so')
BITS = (0,) if head else (0, 1, 1024, 64)
import statistics
times = {(n, bits): [] for n in cases for bits in BITS}
for rnd in range(9):                                 # round-robin over every (case, variant): clock drift hits all alike; medians
    order = list(times)
    if rnd % 2:
        order.reverse()
    for n, bits in order:
        ops.debug_set(2, bits)
        cases[n]()
        times[(n, bits)].append(t(cases[n]))
ops.debug_set(2, 0)
print('| case | normal us | no DMA | no reads/MFMA | no epilogue |')
print('|---|---|---|---|---|')
for n in cases:
    print(f'| {n} | ' + ' | '.join(f'{statistics.median(times[(n, bits)][1:]):.0f}' for bits in BITS) + ' |')
