"""Micro-benchmark of single conv layers through libodtk (GPU), legacy vs 8-wave engine.

usage: python tools/conv_bench.py [layers|all] [passes] [reps] [modes]
  modes: comma list of odtk_debug_set(1, mode) values (1 = legacy, 2 = 8-wave v3, 3 = persistent v4, 0 = auto);
         1xx / 2xx = v3 / v4 with perf-experiment bits xx; Ebbb (>= 1000) / Ebbbb (>= 10000) / Ebbbbbb (>= 1000000) = engine E with bits
"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from odtk import ops

LAYERS = {  # name: (N, H, C, K, k, stride, dil)   (C = padded input channels)
    'conv1_1': (32, 300, 8, 64, 3, 1, 1),
    'conv1_2': (32, 300, 64, 64, 3, 1, 1),
    'conv1_2s': (4, 300, 64, 64, 3, 1, 1),
    'conv2_1': (32, 150, 64, 128, 3, 1, 1),
    'conv2_2': (32, 150, 128, 128, 3, 1, 1),
    'conv3_1': (32, 75, 128, 256, 3, 1, 1),
    'conv3_2': (32, 75, 256, 256, 3, 1, 1),
    'conv4_1': (32, 38, 256, 512, 3, 1, 1),
    'conv4_2': (32, 38, 512, 512, 3, 1, 1),
    'conv5_2': (32, 19, 512, 512, 3, 1, 1),
    'conv6': (32, 19, 512, 1024, 3, 1, 2),
    'conv7': (32, 19, 1024, 1024, 1, 1, 1),
    'conv8_1': (32, 19, 1024, 256, 1, 1, 1),
    'conv8_2': (32, 19, 256, 512, 3, 2, 1),
    'pred1': (32, 38, 512, 100, 3, 1, 1),
    'pred2': (32, 19, 1024, 150, 3, 1, 1),
    'pred3': (32, 10, 512, 150, 3, 1, 1),
    'pred4': (32, 5, 256, 150, 3, 1, 1),
    'pred5': (32, 5, 256, 100, 3, 1, 1),
    'pred6': (32, 3, 256, 100, 3, 1, 1),
    'conv9_1': (32, 10, 512, 128, 1, 1, 1),
    'conv9_2': (32, 10, 128, 256, 3, 2, 1),
    'conv10_1': (32, 5, 256, 128, 1, 1, 1),
    'conv10_2': (32, 5, 128, 256, 3, 1, 1),
    'conv10_2s': (32, 5, 128, 256, 3, 2, 1),
    # DarkNet-53 at 8 images (BASELINE config 4's per-GPU share)
    'y13_3': (8, 13, 512, 1024, 3, 1, 1), 'y13_1': (8, 13, 1024, 512, 1, 1, 1), 'y26_3': (8, 26, 256, 512, 3, 1, 1), 'y26_1': (8, 26, 512, 256, 1, 1, 1),
    'y52_3': (8, 52, 128, 256, 3, 1, 1), 'y52_1': (8, 52, 256, 128, 1, 1, 1), 'y26_s2': (8, 26, 512, 1024, 3, 2, 1), 'y52_s2': (8, 52, 256, 512, 3, 2, 1),
    'f32_3': (16, 32, 256, 256, 3, 1, 1), 'f16_3': (16, 16, 256, 256, 3, 1, 1),      # FCOS 512 x 512 at 16 images: the towers on P4 / P5
    'y104_s2': (8, 104, 128, 256, 3, 2, 1), 'y208_s2': (8, 208, 64, 128, 3, 2, 1), 'y416_s2': (8, 416, 32, 64, 3, 2, 1),
}
which = list(LAYERS) if len(sys.argv) < 2 or sys.argv[1] == 'all' else sys.argv[1].split(',')
passes = sys.argv[2].split(',') if len(sys.argv) > 2 else ['fwd', 'dgrad', 'wgrad']
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
modes = [(m if ':' in m else int(m)) for m in sys.argv[4].split(',')] if len(sys.argv) > 4 else [1, 2]      # 'E:bits' = engine E, debug bits
# mode >= 100: 8-wave kernels with perf-experiment bits (mode - 100) -> odtk_debug_set(2, bits); results are garbage
dev = torch.device('cuda')
if os.environ.get('ODTK_WG'):
    ops.debug_set(5, int(os.environ['ODTK_WG']))            # filter-gradient flush: 0 = float atomics, 1 (default since round 6) = partial stores + reduction launch (include/odtk.h, key 5)
if os.environ.get('ODTK_DBG2'):
    ops.debug_set(6, int(os.environ['ODTK_DBG2']))          # dispatch A/B switches that leave results intact (include/odtk.h, key 6)
tot = {m: 0.0 for m in modes}
for name in which:
    N, H, C, K, k, s, d = LAYERS[name]
    Kp = ops.pad_to(K, int(os.environ.get('KP_PAD', 8)))
    desc = ops.conv_desc(N, H, H, C, C, K, Kp, k, s, d, ops.BF16, ops.BF16)
    M = N * desc.Ho * desc.Wo
    x = torch.randn(N * H * H, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(K, k, k, C, device=dev) * 0.05)
    wc = w.to(torch.bfloat16).contiguous()
    wt = torch.empty(C * k * k * Kp, dtype=torch.bfloat16, device=dev)
    ops.filter_prepare(w, K, k, k, C, Kp, ops.BF16, None, wt)
    y = torch.zeros(M, Kp, dtype=torch.bfloat16, device=dev)
    dy = torch.randn(M, Kp, device=dev).to(torch.bfloat16)
    dx = torch.empty_like(x)
    dw = torch.zeros(K, k, k, C, device=dev)
    bias = torch.zeros(K, device=dev)
    fl = 2.0 * M * K * C * k * k
    fns = {'fwd': lambda: ops.conv2d_fwd(desc, x, wc, bias, y, True),
           'dgrad': lambda: ops.conv2d_dgrad(desc, dy, Kp, wt, x, dx, False),
           'wgrad': lambda: ops.conv2d_wgrad(desc, x, dy, Kp, dw, bias)}
    for p in passes:
        f = fns[p]
        line = f'{name:8s} {p:6s}'
        for mode in modes:
            if isinstance(mode, str):
                ops.debug_set(1, int(mode.split(':')[0])); ops.debug_set(2, int(mode.split(':')[1]))
            elif mode >= 1000000:
                ops.debug_set(1, mode // 1000000); ops.debug_set(2, mode % 1000000)
            else:
              ops.debug_set(1, mode // 10000 if mode >= 10000 else mode // 1000 if mode >= 1000 else 3 if mode >= 200 else 2 if mode >= 100 else mode)
              ops.debug_set(2, mode % 10000 if mode >= 10000 else mode % 1000 if mode >= 1000 else mode % 100 if mode >= 100 else 0)
            for _ in range(3):
                f()
            torch.cuda.synchronize()
            s0 = torch.cuda.Event(enable_timing=True); s1 = torch.cuda.Event(enable_timing=True)
            s0.record()
            for _ in range(reps):
                f()
            s1.record(); torch.cuda.synchronize()
            t = s0.elapsed_time(s1) / reps * 1e-3
            tot[mode] += t
            line += f' | mode{mode} {t*1e6:8.1f} us {fl/t/1e12:7.1f} TF [{ops.conv_last_kernel()}]'
        print(line, flush=True)
ops.debug_set(1, 0)
ops.debug_set(2, 0)
print('sum of listed launches: ' + ', '.join(f'mode{m} {tot[m]*1e3:.3f} ms' for m in modes))
