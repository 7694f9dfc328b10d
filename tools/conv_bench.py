"""Micro-benchmark of single conv layers through libodtk (GPU)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from odtk import ops

LAYERS = {  # name: (N, H, C, K, k, stride, dil)
    'conv1_2': (32, 300, 64, 64, 3, 1, 1),
    'conv2_2': (32, 150, 128, 128, 3, 1, 1),
    'conv3_2': (32, 75, 256, 256, 3, 1, 1),
    'conv4_2': (32, 38, 512, 512, 3, 1, 1),
    'conv5_2': (32, 19, 512, 512, 3, 1, 1),
    'conv6': (32, 19, 512, 1024, 3, 1, 2),
    'conv7': (32, 19, 1024, 1024, 1, 1, 1),
}
which = sys.argv[1].split(',') if len(sys.argv) > 1 else list(LAYERS)
passes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fwd', 'dgrad', 'wgrad']
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device('cuda')
for name in which:
    N, H, C, K, k, s, d = LAYERS[name]
    desc = ops.conv_desc(N, H, H, C, C, K, K, k, s, d, ops.BF16, ops.BF16)
    M = N * desc.Ho * desc.Wo
    x = torch.randn(N * H * H, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(K, k, k, C, device=dev) * 0.05)
    wc = w.to(torch.bfloat16).contiguous()
    wt = torch.empty(C * k * k * K, dtype=torch.bfloat16, device=dev)
    ops.filter_prepare(w, K, k, k, C, K, ops.BF16, None, wt)
    y = torch.empty(M, K, dtype=torch.bfloat16, device=dev)
    dy = torch.randn(M, K, device=dev).to(torch.bfloat16)
    dx = torch.empty_like(x)
    dw = torch.zeros(K, k, k, C, device=dev)
    bias = torch.zeros(K, device=dev)
    fl = 2.0 * M * K * C * k * k
    fns = {'fwd': lambda: ops.conv2d_fwd(desc, x, wc, bias, y, True),
           'dgrad': lambda: ops.conv2d_dgrad(desc, dy, K, wt, x, dx, False),
           'wgrad': lambda: ops.conv2d_wgrad(desc, x, dy, K, dw, bias)}
    for p in passes:
        f = fns[p]
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(reps):
            f()
        s1.record(); torch.cuda.synchronize()
        t = s0.elapsed_time(s1) / reps * 1e-3
        print(f'{name:8s} {p:6s} {t*1e6:9.1f} us  {fl/t/1e12:8.1f} TFLOP/s', flush=True)
