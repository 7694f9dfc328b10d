"""YOLOv3 DarkNet-53 training throughput at BASELINE config 4's per-GPU size (416 x 416, batch 64 over 8 GPUs = 8 per GPU), and at
larger per-GPU batches.  Synthetic VOC-shaped batch, random-init weights, bf16 engine, full step (forward, loss, backward, optimizer).
usage: python tools/yolov3_bench.py [batch=8] [steps=10] [size=416]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import odtk
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synth as S

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
size = int(sys.argv[3]) if len(sys.argv) > 3 else 416
cfg = {'mode': 'train', 'data_shape': [size, size, 3], 'num_classes': 20, 'weight_decay': 5e-4, 'keep_prob': 0.5, 'data_format': 'channels_last',
       'batch_size': batch, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'num_priors': 3, 'nms_score_threshold': 0.5,
       'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'verbose': False, 'compute_dtype': os.environ.get('ODTK_ENGINE', 'bf16'),      # (the class default is f32x3 since round 6)
       'priors': [[[10., 13.], [16, 30.], [33., 23.]], [[30., 61.], [62., 45.], [59., 119.]], [[116., 90.], [156., 198.], [373., 326.]]]}
g = torch.Generator().manual_seed(0)
imgs = (torch.rand(batch, size, size, 3, generator=g) * 255).round()
gt = S.synthetic_gt(batch, size, 1, lo=0.05, hi=0.8)
if os.environ.get('ODTK_DBG'):
    from odtk import ops as _ops
    _ops.debug_set(2, int(os.environ['ODTK_DBG']))
m = odtk.YOLOv3(cfg, {'num_train': batch, 'train_generator': [(imgs, gt)], 'val_generator': None, 'num_val': 0})
m.set_batch(imgs, gt)
for _ in range(3):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = m.train_step(1e-4)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
flops = 0
for name, cin, cout, k, s, _ in m.specs:
    d = m.desc[name]
    flops += 2 * batch * d.Ho * d.Wo * cout * cin * k * k
flops *= 3                                   # forward + dgrad + wgrad (the first layer has no dgrad: < 0.1 %)
print(f'YOLOv3 {size}x{size} batch {batch} bf16: {dt * 1e3:8.2f} ms/step  {batch / dt:8.1f} images/s   conv {flops / dt / 1e12:6.1f} TFLOP/s   loss {float(loss):.3f}')
if os.environ.get('YOLO_TABLE'):
    # per-layer conv launches with HIP events (bench.py's ConvTimer), grouped by shape
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from odtk import ops
    timer = bench.ConvTimer(ops)
    timer.install()
    timer.enabled = True
    nst = 5
    for _ in range(nst):
        m.train_step(1e-4)
    torch.cuda.synchronize()
    timer.enabled = False
    rows = {}
    for kind, kern, fl, tt, ab, shp in timer.per_step(nst):
        r = rows.setdefault((kind, kern, shp), [0, 0.0, 0.0]); r[0] += 1; r[1] += tt; r[2] += fl
    tot = sum(r[1] for r in rows.values())
    print(f'conv launches per step: {sum(r[0] for r in rows.values())}, {tot * 1e3:.2f} ms')
    for (kind, kern, shp), r in sorted(rows.items(), key=lambda kv: -kv[1][1])[:45]:
        N, H, W, C_, K, R, s, dl = shp
        print(f'{kind:13s} {kern:38s} H{H:<3d} C{C_:<4d} K{K:<4d} k{R} s{s}  x{r[0]:<2d} {r[1] * 1e6 / r[0]:7.1f} us each {r[1] * 1e3:6.3f} ms  {r[2] / r[1] / 1e12:6.1f} TF')
if os.environ.get('YOLO_PROFILE'):
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(2):
            m.train_step(1e-4)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=14, max_name_column_width=60))
