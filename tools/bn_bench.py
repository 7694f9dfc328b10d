"""batch norm forward / backward of the SSD300 layers that carry one (conv6 .. conv11_2, the heads) at batch 32, isolated (GPU): us per call and the
bytes it has to move.   python tools/bn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
from odtk import ops
dev = torch.device('cuda')
SHAPES = {'conv6/conv7': (32 * 19 * 19, 1024, True), 'conv8_1': (32 * 19 * 19, 256, True), 'conv8_2': (3200, 512, True), 'conv9_1': (3200, 128, True),
          'conv9_2': (800, 256, True), 'conv11_2': (288, 256, True), 'pred1': (32 * 38 * 38, 100, False), 'pred2': (32 * 19 * 19, 150, False)}
if len(sys.argv) > 1 and sys.argv[1] == 'big':       # the large maps of the other configurations: DarkNet-53 at 416 x 416 x 8 (bf16, leaky ReLU), RetinaNet / CenterNet-like f32
    SHAPES = {'y416x32': (8 * 416 * 416, 32, True), 'y208x64': (8 * 208 * 208, 64, True), 'y208x32': (8 * 208 * 208, 32, True), 'y104x128': (8 * 104 * 104, 128, True),
              'y104x64': (8 * 104 * 104, 64, True), 'y52x256': (8 * 52 * 52, 256, True), 'y52x128': (8 * 52 * 52, 128, True), 'y26x512': (8 * 26 * 26, 512, True)}
if len(sys.argv) > 1 and sys.argv[1] == 'narrow':    # DLA-34 (CenterNet 512 x 512 at 16 images) and DarkNet-53's first layers: the largest maps, few channels
    SHAPES = {'c512x16': (16 * 512 * 512, 16, True), 'c256x32': (16 * 256 * 256, 32, True), 'c128x64': (16 * 128 * 128, 64, True), 'c128x128': (16 * 128 * 128, 128, True),
              'c64x256': (16 * 64 * 64, 256, True), 'y416x32': (8 * 416 * 416, 32, True), 'y208x64': (8 * 208 * 208, 64, True), 'y52x256': (8 * 52 * 52, 256, True)}
if os.environ.get('ODTK_BN_RPB'):
    ops.debug_set(4, int(os.environ['ODTK_BN_RPB']))      # -20 .. -23: 128 / 512 / 1 024 / 256 rows per workgroup of the apply passes (A/B)
def timeit(f, n=30):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, (M, C, relu) in SHAPES.items():
    ld = ops.pad_to(C, 8)
    z = torch.randn(M, ld, device=dev).to(torch.bfloat16)
    ydt = torch.bfloat16 if relu else torch.float32
    y = torch.zeros(M, C if not relu else ld, dtype=ydt, device=dev)
    dy = torch.randn_like(y)
    dz = torch.zeros_like(z)
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mm, mv, sm, si = torch.zeros(C, device=dev), torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    ws = torch.zeros(ops.bn_workspace_bytes(M, C), dtype=torch.uint8, device=dev)
    ldy = y.shape[1]
    tf = timeit(lambda: ops.bn_fwd(z, M, C, ld, g, b, mm, mv, sm, si, True, relu, y, ldy, M, 0, ws))
    tb = timeit(lambda: ops.bn_bwd(z, y if relu else None, dy, M, C, ld, ldy, M, 0, g, sm, si, relu, dz, dg, db, ws))
    zb = M * ld * 2; yb = M * ldy * y.element_size()
    fwd_bytes = 2 * zb + yb; bwd_bytes = 2 * (zb + yb + (yb if relu else 0)) + zb
    print(f'{name:12s} M={M:6d} C={C:5d}  fwd {tf:6.1f} us ({fwd_bytes / tf / 1e6:5.2f} TB/s over {fwd_bytes / 1e6:5.1f} MB)   bwd {tb:6.1f} us ({bwd_bytes / tb / 1e6:5.2f} TB/s over {bwd_bytes / 1e6:5.1f} MB)', flush=True)
