"""bf16 RetinaNet gradients vs the f32 oracle, layer by layer in backward order.  usage (GPU box): python tests/tools/debug_retina_bf16.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import retinanet_net_ref as NR
import test_gpu_retinanet_model as T

torch.set_num_threads(16)
dt = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
p = NR.init_params(11)
batch = T._batch(2, 160, 100)
m = T._model('train', dt, 2, 160, T._provider([batch]))
m.load_oracle_params(p)
m.set_batch(*batch)
m.G.zero_(); m._forward(True); m._loss(0.5)
for _ in m._backward_iter():
    pass
torch.cuda.synchronize()
q = {k: v.clone() for k, v in p.items()}
mom = {k: torch.zeros_like(v) for k, v in p.items() if k in NR.trainable_names(p)}
total, data, grads = NR.train_step(q, mom, batch[0], batch[1], 0.005)
cos = lambda a, b: float(torch.dot(a.reshape(-1), b.reshape(-1)) / (a.norm() * b.norm() + 1e-20))
for i in list(range(121, 60, -1)) + [50, 30, 10, 4, 3, 2, 1, 0]:
    k = f'l{i}.w'
    a, b = m.get_param(k, m.G), grads[k] - 1e-4 * p[k]
    print(k, tuple(a.shape), 'cos %.4f' % cos(a, b), 'norm ratio %.3f' % (float(a.norm()) / (float(b.norm()) + 1e-20)), 'gamma cos %.4f' % cos(m.get_param(f'l{i}.gamma', m.G), grads[f'l{i}.gamma']))
