"""CPU, world_size 2 (gloo): the bucketed gradient all-reduce used for data-parallel training."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import odtk  # noqa: F401
    from odtk.dist import BucketAllReducer
    sizes = [64, 1000, 64, 50000, 128, 30000, 64, 7000, 192]
    segs, off = [], 0
    for i, n in enumerate(sizes):
        segs.append((f'l{i}', off, off + n)); off += n
    flat = torch.zeros(off)
    red = BucketAllReducer(flat, segs, None, bucket_bytes=100_000)
    # buckets tile the buffer exactly once, suffix first
    cover = sorted((s, e) for s, e, _ in red.buckets)
    assert cover[0][0] == 0 and cover[-1][1] == off and all(a[1] == b[0] for a, b in zip(cover, cover[1:]))
    assert red.buckets[0][1] == off
    ok = True
    for step in range(3):
        flat.zero_()
        red.begin_step()
        for i in reversed(range(len(segs))):          # backward order: last layer first
            name, s, e = segs[i]
            flat[s:e] = torch.arange(s, e, dtype=torch.float32) * (rank + 1) + step
            red.segment_ready(name)
        red.finish_step()
        exp = torch.arange(0, off, dtype=torch.float32) * sum(r + 1 for r in range(world)) + step * world
        ok = ok and torch.allclose(flat, exp)
    q.put((rank, ok, len(red.buckets)))
    dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    assert res[0][2] >= 3          # several buckets -> overlap opportunities


def _hook_worker(rank, world, port, q):
    """BucketAllReducer's `collective=` hook (what OdtkCollective plugs into on the GPU) in a world of TWO ranks: an object with all_reduce(buf) -> handle
    stands in for the C-ABI collective (here over gloo), for f32 and bf16 buckets; the sums and the launch order equal the torch.distributed path's."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import odtk  # noqa: F401
    from odtk.dist import BucketAllReducer

    class Handle:
        def __init__(self, work, log): self.work, self.log = work, log
        def wait(self): self.work.wait(); self.log.append('wait')

    class Loopback:                                       # the interface of odtk.dist.OdtkCollective
        def __init__(self): self.calls, self.log = [], []
        def all_reduce(self, buf):
            self.calls.append((buf.data_ptr(), buf.numel(), buf.dtype))
            return Handle(dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True), self.log)

    sizes = [64, 1000, 64, 50000, 128, 30000, 64, 7000, 192]
    segs, off = [], 0
    for i, n in enumerate(sizes):
        segs.append((f'l{i}', off, off + n)); off += n
    ok = True
    for comm_dtype in ('f32', 'bf16'):
        outs = []
        for coll in (None, Loopback()):
            flat = torch.zeros(off)
            red = BucketAllReducer(flat, segs, None, bucket_bytes=100_000, comm_dtype=comm_dtype, collective=coll)
            red.begin_step()
            for i in reversed(range(len(segs))):
                name, s, e = segs[i]
                flat[s:e] = torch.arange(s, e, dtype=torch.float32) * 0.001 * (rank + 1)
                red.segment_ready(name)
            red.finish_step()
            outs.append((flat.clone(), list(red.launch_log)))
            if coll is not None:
                ok = ok and len(coll.calls) == len(red.buckets) and coll.log == ['wait'] * len(red.buckets)
                ok = ok and all(dt == (torch.float32 if comm_dtype == 'f32' else torch.bfloat16) for _, _, dt in coll.calls)
                ok = ok and [n for _, n, _ in coll.calls] == [e - s for s, e in red.launch_log]
        ok = ok and torch.equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]
        exp = torch.arange(0, off, dtype=torch.float32) * 0.001 * 3
        ok = ok and torch.allclose(outs[1][0], exp, rtol=2e-2 if comm_dtype == 'bf16' else 1e-6, atol=1e-3 if comm_dtype == 'bf16' else 1e-6)
    q.put((rank, ok, 0))
    dist.destroy_process_group()


def test_bucket_reducer_collective_hook_world2():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hook_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res), res


def test_single_process_is_a_noop():
    import odtk  # noqa: F401
    from odtk.dist import BucketAllReducer
    flat = torch.arange(100, dtype=torch.float32)
    red = BucketAllReducer(flat, [('a', 0, 40), ('b', 40, 100)], None, bucket_bytes=64)
    red.begin_step(); red.segment_ready('b'); red.segment_ready('a'); red.finish_step()
    assert torch.equal(flat, torch.arange(100, dtype=torch.float32))


def _yolo_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    import types
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import odtk  # noqa: F401
    from odtk.dist import GradAllReducer
    from odtk.yolov3 import YOLOv3, layer_specs
    specs = layer_specs(20, 3)
    pinfo, nparam, _, _ = YOLOv3.param_layout(specs, 8)
    model = types.SimpleNamespace(pinfo=pinfo, nparam=nparam, G=torch.zeros(nparam))
    red = GradAllReducer(model, None, bucket_mb=25)
    layers = [s[0] for s in specs]
    launched = []
    ok = True
    for step in range(2):
        model.G.zero_()
        red.begin_step()
        for name in reversed(layers):                       # YOLOv3._backward_iter finishes c74 first, c0 last
            for suffix in ('.w', '.gamma', '.beta'):
                off, shape = pinfo[name + suffix]
                n = 1
                for d in shape:
                    n *= d
                model.G[off:off + n] = float(rank + 1) * (1 + step)
            before = red.red.next_bucket
            red.layer_ready(name)
            if red.red.next_bucket > before:
                launched.append(name)
        red.finish_step()
        off, shape = pinfo['c30.w']
        ok = ok and float(model.G[off]) == 3.0 * (1 + step) and float(model.G[pinfo['c74.beta'][0]]) == 3.0 * (1 + step)
        ok = ok and float(model.G[pinfo['c0.b'][0]]) == 0.0             # the bias slots carry no gradient
    q.put((rank, ok, len(red.red.buckets), launched[: len(red.red.buckets)], red.boundary_layers()))
    dist.destroy_process_group()


def test_yolov3_gradient_buckets_world2():
    """the data-parallel hooks of the YOLOv3 class on its real parameter layout (62 M parameters, 75 layers), two gloo ranks on
    the CPU: buckets of ~25 MB close in backward order c74 -> c0, each as soon as its lowest layer is done, and the exchanged
    buffer holds the sum of the replicas"""
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_yolo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _, _, _ in res)
    _, _, nb, launched, boundary = res[0]
    assert 8 <= nb <= 12                                    # 247 MB of f32 gradients in ~25 MB buckets
    assert launched == boundary                             # a bucket's all-reduce starts at the layer that closes it
    idx = [int(n[1:]) for n in boundary]
    assert idx == sorted(idx, reverse=True) and idx[-1] == 0


def _yolo_model_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import mock_ops
    import odtk
    torch.set_num_threads(4)
    m, batch = _yolo_cpu_model(rank)
    with mock_ops.installed():
        m.attach_data_parallel(bucket_mb=16)
        m.set_batch(*batch)
        loss = float(m.train_step(0.002))
    torch.save({'P': m.P.clone(), 'G': m.G.clone(), 'loss': loss}, os.path.join(out_dir, f'm{rank}.pt'))
    dist.destroy_process_group()


def _yolo_cpu_model(rank):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import mock_ops
    import odtk
    from oracle import yolov3_ref as YR
    cfg = {'mode': 'train', 'data_shape': [64, 64, 3], 'num_classes': 20, 'weight_decay': 5e-4, 'keep_prob': 0.5, 'data_format': 'channels_last',
           'batch_size': 1, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1., 'num_priors': 3, 'nms_score_threshold': 0.5,
           'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'priors': YR.PRIORS_PX, 'verbose': False, 'compute_dtype': 'f32', 'device': 'cpu',
           'use_graph': False, 'seed': 3}
    g = torch.Generator().manual_seed(800 + rank)
    batch = ((torch.rand(1, 64, 64, 3, generator=g) * 255).round(), YR.synthetic_gt(1, 64, 850 + rank, max_obj=3))
    with mock_ops.installed():
        m = odtk.YOLOv3(cfg, {'num_train': 1, 'train_generator': [], 'val_generator': None, 'num_val': 0})
    return m, batch


def test_yolov3_model_data_parallel_world2_on_cpu(tmp_path):
    """the whole data-parallel training step of the YOLOv3 class on two gloo ranks, every libodtk launch mocked on the CPU: replicas end
    identical, and the exchanged gradient is the sum of the ranks' local gradients with the loss divided by the GLOBAL batch"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mock_ops
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_yolo_model_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
        assert p.exitcode == 0
    a, b = torch.load(os.path.join(tmp_path, 'm0.pt')), torch.load(os.path.join(tmp_path, 'm1.pt'))
    assert torch.equal(a['P'], b['P']) and torch.equal(a['G'], b['G'])
    total = None
    threads = torch.get_num_threads()
    torch.set_num_threads(4)                     # as in the workers: the same reduction order inside the CPU convolutions
    for rank in range(2):
        m, batch = _yolo_cpu_model(rank)
        with mock_ops.installed():
            m.set_batch(*batch)
            m.G.zero_(); m._forward(True); m._loss(0.5 / 2)
            for _ in m._backward_iter():
                pass
        total = m.G.clone() if total is None else total + m.G
    torch.set_num_threads(threads)
    # batch 1 at 64 x 64 puts batch norms over 4 samples into the model: a leaky-ReLU input that lands on the other side of 0 in one of the
    # two computations moves the gradient by per cents (tests/test_gpu_yolov3.py); identical thread counts usually give identical bits
    assert float((a['G'] - total).norm()) < 5e-2 * float(total.norm())


# ---------------------------------------------------------------------------------------------------------------------------------
# the shared graph engine of refinedet.py (RefineDet320, PFPNetR, YOLOv2): data-parallel hooks on two gloo ranks, launches mocked
def _engine_cpu_model(kind, rank):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import mock_ops
    import odtk
    g = torch.Generator().manual_seed(900 + rank)
    if kind == 'yolov2':
        from oracle import yolov2_ref as YR
        cfg = {'mode': 'train', 'is_pretraining': False, 'data_shape': [128, 160, 3], 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5,
               'data_format': 'channels_last', 'batch_size': 1, 'coord_scale': 1, 'noobj_scale': 1, 'obj_scale': 5., 'class_scale': 1.,
               'nms_score_threshold': 0.5, 'nms_max_boxes': 10, 'nms_iou_threshold': 0.5, 'rescore_confidence': False, 'priors': YR.PRIORS, 'verbose': False,
               'compute_dtype': 'f32', 'device': 'cpu', 'seed': 3}
        batch = ((torch.rand(1, 128, 160, 3, generator=g) * 255).round(), YR.synthetic_gt(1, 128, 950 + rank, pad=6, max_obj=3))
        with mock_ops.installed():
            m = odtk.YOLOv2(cfg, {'data_shape': [128, 160, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    elif kind == 'lhrcnn':
        from oracle import lhrcnn_ref as LR
        cfg = {'data_shape': [224, 288, 3], 'mode': 'train', 'is_pretraining': False, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4,
               'keep_prob': 0.5, 'batch_size': 1, 'rpn_first_step': 60000, 'rcnn_first_step': 100000, 'rpn_second_step': 160000, 'nms_score_threshold': 0.5,
               'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'post_nms_proposal': 500, 'verbose': False, 'device': 'cpu', 'seed': 3}
        batch = ((torch.rand(1, 224, 288, 3, generator=g) * 255).round(), LR.synthetic_gt(1, 224, 288, 950 + rank))
        with mock_ops.installed():
            m = odtk.LHRCNN(cfg, {'data_shape': [224, 288, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    else:
        from oracle import refinedet_ref as FR
        cfg = {'mode': 'train', 'input_size': 320, 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
               'nms_score_threshold': 0.1, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.45, 'pretraining_weight': '', 'verbose': False, 'compute_dtype': 'f32',
               'device': 'cpu', 'seed': 3}
        batch = ((torch.rand(1, 320, 320, 3, generator=g) * 255).round(), FR.synthetic_gt(1, 320, 950 + rank, pad=8, max_obj=3))
        with mock_ops.installed():
            m = odtk.PFPNetR(cfg, {'data_shape': [320, 320, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
    return m, batch


def _engine_model_worker(kind, rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import mock_ops
    torch.set_num_threads(4)
    m, batch = _engine_cpu_model(kind, rank)
    with mock_ops.installed():
        red = m.attach_data_parallel(bucket_mb=16)
        m.set_batch(*batch)
        loss = float(m.train_step(0.002))
    torch.save({'P': m.P.clone(), 'G': m.G.clone(), 'loss': loss, 'buckets': len(red.red.buckets)}, os.path.join(out_dir, f'{kind}{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["yolov2", "pfpnet", "lhrcnn"])
def test_engine_models_data_parallel_world2_on_cpu(kind, tmp_path):
    """YOLOv2, PFPNetR and LHRCNN (the latter: the R-CNN stage outside the plan reports the dense layers first, a separable layer is ready behind its depthwise
    entry) (refinedet.py's engine: per-layer readiness reported while the backward plan runs, L2-norm scalars and multi-consumer
    gradient buffers included) on two gloo ranks with every libodtk launch mocked: replicas end identical, several buckets were exchanged, and the
    exchanged gradient is the sum of the ranks' local gradients with the loss divided by the GLOBAL batch"""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import mock_ops
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_engine_model_worker, args=(kind, r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=900)
        assert p.exitcode == 0
    a, b = torch.load(os.path.join(tmp_path, f'{kind}0.pt')), torch.load(os.path.join(tmp_path, f'{kind}1.pt'))
    assert torch.equal(a['P'], b['P']) and torch.equal(a['G'], b['G']) and a['buckets'] >= 3
    total = None
    threads = torch.get_num_threads()
    torch.set_num_threads(4)
    for rank in range(2):
        m, batch = _engine_cpu_model(kind, rank)
        with mock_ops.installed():
            m.loss_divisor_batch = 2                          # what attach_data_parallel sets: the global batch
            m.set_batch(*batch)
            m._step_body()
        total = m.G.clone() if total is None else total + m.G
    torch.set_num_threads(threads)
    assert float((a['G'] - total).norm()) < 5e-2 * float(total.norm())


def _bf16_worker(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from odtk.dist import BucketAllReducer
    sizes = [64, 1024, 64, 50048, 128]
    segs, off = [], 0
    for i, n in enumerate(sizes):
        segs.append((f'l{i}', off, off + n)); off += n
    g = torch.Generator().manual_seed(10 + rank)
    mine = torch.randn(off, generator=g)
    others = [torch.randn(off, generator=torch.Generator().manual_seed(10 + r)) for r in range(world)]
    flat = torch.zeros(off)
    red = BucketAllReducer(flat, segs, None, bucket_bytes=100_000, comm_dtype='bf16', force_collectives=(world == 1))
    ok = len(red.buckets) >= 2
    for step in range(2):
        flat.copy_(mine)
        red.begin_step()
        for name, _, _ in reversed(segs):
            red.segment_ready(name)
        red.finish_step()
        # what the wire format gives: every rank's contribution rounded to bf16, summed (gloo sums bf16 pairwise; one more rounding)
        exact = sum(o.to(torch.bfloat16).float() for o in others)
        ok = ok and float((flat - exact).abs().max()) <= 2 ** -7 * float(exact.abs().max())
        ok = ok and float((flat - sum(others)).abs().max()) <= 3 * 2 ** -8 * float(sum(o.abs() for o in others).max())
    q.put((rank, ok, red.comm_dtype))
    dist.destroy_process_group()


@pytest.mark.parametrize('world', [1, 2])
def test_bf16_gradient_buckets(world):
    """comm_dtype='bf16': the buckets are narrowed, summed by the collective as bf16 and widened back; world 1 runs with
    force_collectives (the single-GPU RCCL exercise of `bench.py --dp-world1`, here on gloo)"""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bf16_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    assert all(ok for _, ok, _ in res) and all(dt == 'bf16' for _, _, dt in res), res


# ---------------------------------------------------------------------------------------------------------------------------------
# SSD300 itself: the data-parallel step keeps the streams of the single-device step (round 4)
def _ssd300_dp_worker(rank, world, port, out_dir):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import contextlib
    import sys
    from unittest import mock
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here)); sys.path.insert(0, here)
    import mock_ops
    import odtk
    from oracle import ssd300_ref as R
    torch.set_num_threads(4)
    log = []

    class FakeStream:                                     # the CPU executes in program order: a stream is a label, a wait is a log line
        def __init__(self, name):
            self.name = name

        def wait_stream(self, other):
            log.append(('wait', self.name, other.name))

        def wait_event(self, ev):
            log.append(('wait', self.name, ev.on))
            log.append(('wait_event', self.name, ev.seq))

    class FakeEvent:
        on = None
        seq = None

        def __init__(self, *a, **k):
            pass

        def record(self, st):
            self.on = st.name
            self.seq = len(log)
            log.append(('record', st.name, self.seq))

    main = FakeStream('main')
    cur = [main]

    @contextlib.contextmanager
    def on_stream(s):
        prev, cur[0] = cur[0], s
        try:
            yield
        finally:
            cur[0] = prev
    cfg = {'mode': 'train', 'data_format': 'channels_last', 'num_classes': 20, 'weight_decay': 1e-4, 'keep_prob': 0.5, 'batch_size': 1,
           'nms_score_threshold': 0.5, 'nms_max_boxes': 20, 'nms_iou_threshold': 0.5, 'pretraining_weight': '', 'verbose': False,
           'compute_dtype': 'f32', 'seed': 0, 'use_graph': False, 'device': 'cpu'}
    imgs, gt = R.synthetic_batch(1, 40 + rank)
    real_all_reduce = dist.all_reduce

    def spy(buf, **kw):
        log.append(('all_reduce', cur[0].name, int(buf.numel())))
        return real_all_reduce(buf, **kw)
    out = {}
    for mode in ('streams', 'single'):
        del log[:]
        with mock_ops.installed(), mock.patch.object(torch.cuda, 'current_stream', lambda *a: cur[0]), mock.patch.object(torch.cuda, 'stream', on_stream), \
                mock.patch.object(torch.cuda, 'Event', FakeEvent), mock.patch.object(dist, 'all_reduce', spy):
            for fname in ('conv2d_dgrad', 'conv2d_dgrad_bits'):           # where the input-gradient launches sit between the records, waits and collectives
                def logged(*a, _f=getattr(odtk.ops, fname), **k):
                    log.append(('dgrad', cur[0].name))
                    return _f(*a, **k)
                setattr(odtk.ops, fname, logged)
            m = odtk.SSD300(cfg, {'data_shape': [300, 300, 3], 'num_train': 1, 'num_val': 0, 'train_generator': [], 'val_generator': None})
            if mode == 'streams':                          # what the GPU build has: head stream + tail filter-gradient stream
                m._tail, m._twg = FakeStream('tail'), FakeStream('twg')
                m.ws_tail = torch.zeros_like(m.ws)
            red = m.attach_data_parallel(bucket_mb=8)
            marks = []
            real_ready = red.layer_ready
            red.layer_ready = lambda name: (marks.append(name), real_ready(name))[1]
            m.set_batch(imgs, gt)
            loss = float(m.train_step(0.01))
        out[mode] = {'G': m.G.clone(), 'P': m.P.clone(), 'loss': loss, 'order': list(red.red.launch_log), 'marks': marks, 'log': list(log),
                     'index': dict(red.red.index), 'nbuckets': len(red.red.buckets)}
    torch.save(out, os.path.join(out_dir, f'ssd300_{rank}.pt'))
    dist.destroy_process_group()


def test_ssd300_data_parallel_step_keeps_its_streams_world2(tmp_path):
    """SSD300 on two gloo ranks, every libodtk launch mocked, with the head stream and the tail filter-gradient stream present (labels on the CPU) and without:
    the readiness marks arrive suffix-first in both, the buckets close in the SAME order, every collective is launched from the filter-gradient stream after
    that stream was told to wait for the main and head streams (the main chain never waits for a collective before the optimizer), and parameters / gradients
    after the step are identical in both modes and on both ranks"""
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_ssd300_dp_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=1200)
        assert p.exitcode == 0
    a, b = torch.load(os.path.join(tmp_path, 'ssd300_0.pt')), torch.load(os.path.join(tmp_path, 'ssd300_1.pt'))
    for r in (a, b):
        s, o = r['streams'], r['single']
        assert s['nbuckets'] >= 3 and len(s['order']) == s['nbuckets']
        assert s['order'] == o['order']                                   # bucket launch order: unchanged by the streams
        for mode in (s, o):
            idx = [mode['index'][n] for n in mode['marks']]
            assert idx == sorted(idx, reverse=True) and len(set(idx)) == len(idx) == len(mode['index'])      # every layer once, suffix-first
            assert mode['order'] == sorted(mode['order'], reverse=True)
        assert s['marks'] == o['marks']
        assert torch.equal(s['G'], o['G']) and torch.equal(s['P'], o['P']) and s['loss'] == o['loss']
        # where the collectives were launched from, and what that stream waited for right before
        ar = [i for i, e in enumerate(s['log']) if e[0] == 'all_reduce']
        assert len(ar) == s['nbuckets'] and all(s['log'][i][1] == 'twg' for i in ar)
        behind_dgrad = 0
        for i in ar:
            win = s['log'][max(0, i - 4):i]
            assert ('wait', 'twg', 'main') in win and ('wait', 'twg', 'tail') in win
            # round 6: the main chain is waited for through a per-layer EVENT recorded behind the closing layer's last gradient launch and IN FRONT OF its
            # input-gradient launch -- the collective's stream does not wait for that input gradient (ssd300._dp_grad_point)
            we = [e for e in win if e[0] == 'wait_event' and e[1] == 'twg']
            if we:
                seq = we[-1][2]
                assert s['log'][seq] == ('record', 'main', seq)
                if ('dgrad', 'main') in s['log'][seq:i]:                            # enqueued after the record, before the collective: not waited for
                    behind_dgrad += 1
        # (all but a bucket that closes on a head layer -- nothing on the main chain to point at -- and the last one: conv1_1 has no input gradient)
        assert behind_dgrad >= max(1, len(ar) - 2), (behind_dgrad, len(ar))
        assert not any(e[0] == 'wait' and e[1] == 'main' and e[2] == 'twg' for e in s['log'][:ar[0]])
        assert all(e[1] == 'main' for e in o['log'] if e[0] == 'all_reduce')
    assert torch.equal(a['streams']['P'], b['streams']['P']) and torch.equal(a['streams']['G'], b['streams']['G'])
