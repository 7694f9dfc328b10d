"""Which layer's filter gradient differs when wgrad runs on the second stream?  (debug)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import odtk
from oracle import ssd300_ref as R      # synthetic batch generator only

B = 32
cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
       'batch_size': B, 'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '',
       'verbose': False, 'compute_dtype': 'bf16', 'seed': 0, 'use_graph': len(sys.argv) > 1 and sys.argv[1] == 'graph'}
imgs, gt = R.synthetic_batch(B, 5)
prov = {'data_shape': [300, 300, 3], 'num_train': B, 'num_val': 0, 'train_generator': [], 'val_generator': None}
res = {}
if len(sys.argv) > 2:
    from odtk import ops as _o; _o.debug_set(2, int(sys.argv[2]))
for side in ((False,) if len(sys.argv) > 3 else (False, True)):
    cfg['wgrad_stream'] = side
    m = odtk.SSD300(cfg, prov)
    m.set_batch(imgs, gt)
    worst = {}
    for step in range(6):
        if len(sys.argv) > 4 and sys.argv[4] == 'zeroeager' and step >= 2:
            m._zero_in_graph = False
        r = m.train_step(0.0)                    # lr 0: identical weights every step -> identical gradients
        if len(sys.argv) > 4 and sys.argv[4] == 'keepret':
            res[('ret', step)] = r
        if len(sys.argv) > 4 and sys.argv[4] == 'noback' and m._g_back is not None:
            m._g_back = None
        if len(sys.argv) > 4 and sys.argv[4] == 'nofront' and m._g_front is not None:
            class _F:
                def __init__(s, mm): s.mm = mm
                def replay(s): s.mm._step_front()
            m._g_front = _F(m)
        torch.cuda.synchronize()
        g = m.G.clone()
        mx = float(g.abs().max())
        print('side' if side else 'main', step, '|G|', mx, 'loss', float(m.data_loss), flush=True)
        if mx > 100 or mx != mx:
            badidx = torch.nonzero((g.abs() > 100) | ~torch.isfinite(g)).flatten()
            print('   bad elements', badidx.numel(), 'first', badidx[:6].tolist(), 'last', badidx[-3:].tolist(), 'vals', g[badidx[:4]].tolist())
            print('   |P|', float(m.P.abs().max()), '|Mom|', float(m.Mom.abs().max()), 'G now', float(m.G.abs().max()))
            print('   |dpred|', float(m.dpred.abs().max()), '|pred|', float(m.pred.abs().max()), 'loss_parts', m.loss_parts[0].tolist(), 'cnt', m.m_counts[0].tolist(), 'sel', int(m.sel_cnt[0]))
            for nm in ('pred1', 'pred6', 'conv11_2', 'conv6'):
                sm, si = m.bnsave[nm]
                print('     ', nm, '|z.g|', float(m.zbuf[nm].g.float().abs().max()), '|z.t|', float(m.zbuf[nm].t.float().abs().max()), 'si', float(si.abs().max()), 'sm', float(sm.abs().max()))
            for nm in ('conv5_3', 'conv4_3', 'conv1_2', 'feat1'):
                print('     ', nm, '|g|', float(m.acts[nm].g.float().abs().max()), '|t|', float(m.acts[nm].t.float().abs().max()))
            break
            for name, (off, shape) in m.pinfo.items():
                n = 1
                for s_ in shape: n *= s_
                k = int(((badidx >= off) & (badidx < off + n)).sum())
                if k: print('     ', name, off, n, 'bad', k)
        if step == 0:
            g0 = g
        res[(side, step)] = g
    res[side] = m
ref = res[(False, 0)]
m = res[True] if True in res else res[False]
for step in range(6):
    for side in (False, True):
        g = res[(side, step)]
        bad = []
        for name, (off, shape) in m.pinfo.items():
            n = 1
            for s in shape: n *= s
            a, b = g[off:off + n], ref[off:off + n]
            d = float((a - b).abs().max()); sc = float(b.abs().max()) + 1e-12
            if not (d <= 2e-2 * sc):
                bad.append((name, d / sc))
        print('step', step, 'side' if side else 'main', 'BAD:' if bad else 'ok', bad[:8])
