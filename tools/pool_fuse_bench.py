"""conv + bias + ReLU + 2x2 pool: the fused launch (odtk_conv2d_fwd_pool2x2) against conv then pool, per SSD300 layer at batch 32 (GPU).
    python tools/pool_fuse_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from odtk import ops
dev = torch.device('cuda')
for name, (N, H, C, K) in {'conv1_2+pool1': (32, 300, 64, 64), 'conv2_2+pool2': (32, 150, 128, 128), 'conv3_3+pool3': (32, 75, 256, 256)}.items():
    d = ops.conv_desc(N, H, H, C, C, K, K, 3, 1, 1, ops.BF16, ops.BF16)
    Hp = (H + 1) // 2
    x = torch.randn(N * H * H, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(K, 3, 3, C, device=dev) * 0.05).to(torch.bfloat16).contiguous()
    b = torch.zeros(K, device=dev)
    y = torch.zeros(N * H * H, K, dtype=torch.bfloat16, device=dev)
    p = torch.zeros(N * Hp * Hp, K, dtype=torch.bfloat16, device=dev)
    i = torch.zeros(N * Hp * Hp * K // 8, dtype=torch.int16, device=dev)
    def two():
        ops.conv2d_fwd(d, x, w, b, y, True); ops.maxpool2x2_fwd_idx(y, p, i, N, H, H, K, K, Hp, Hp)
    fns = {'conv then pool': two, 'conv only': lambda: ops.conv2d_fwd(d, x, w, b, y, True),
           'fused, un-pooled map not stored': lambda: ops.conv2d_fwd_pool2x2(d, x, w, b, None, True, p, i),
           'fused, un-pooled map stored too': lambda: ops.conv2d_fwd_pool2x2(d, x, w, b, y, True, p, i)}
    line = f'{name:15s}'
    for k, f in fns.items():
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        line += f' | {k} {e0.elapsed_time(e1) / 20 * 1e3:7.1f} us [{ops.conv_last_kernel()}]'
    print(line, flush=True)
