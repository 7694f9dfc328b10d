"""Data-parallel gradient exchange: one process per GPU, RCCL (torch.distributed backend "nccl")
over xGMI.  The reference is single-device (testSSD300.py:14 pins CUDA_VISIBLE_DEVICES='0'); image
sharding is the only way the path scales, and the only exchange is the gradient sum
(SURVEY.md 8e).

The flat f32 gradient buffer is laid out in forward order, so during backward it becomes final
suffix-first.  It is cut into contiguous buckets; a bucket's all-reduce is launched (async, on
RCCL's own stream) as soon as the lowest layer it covers has finished its wgrad, overlapping
with the rest of backward.  xGMI is point-to-point (7 links/GPU): ~25 MB buckets keep each ring
step large enough to run at link rate while leaving 4-5 buckets to overlap.

Works on CPU tensors with the gloo backend too (tests/test_dist_cpu.py, world_size 2).
"""
from __future__ import annotations

import contextlib
import ctypes

import torch
import torch.distributed as dist

COMM_ID_BYTES = 128          # include/odtk.h: ODTK_COMM_ID_BYTES


class _Done:
    """What BucketAllReducer needs of a collective's handle: wait() orders the CURRENT stream behind it."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class OdtkCollective:
    """The gradient sum through the C-ABI's own collective (include/odtk.h: odtk_comm_*; RCCL underneath) instead of torch.distributed's -- what a
    binder that is not PyTorch would call.  torch.distributed (any backend) is used ONCE, to carry rank 0's 128-byte id to the other ranks; in a
    world of one rank nothing but the library is involved.  all_reduce() has the contract of dist.all_reduce(async_op=True): ordered behind the
    current stream at the call, wait() orders the then-current stream behind it.  It executes on `stream` -- a stream the caller ALREADY owns and
    that is idle during backward (SSD300 / YOLOv3: the box-matching side stream) -- or, without one, on the stream current at the call.  It never
    creates a stream: the HIP runtime deals streams onto four hardware queues, and a fifth one aliases the main chain's (measured: +8.6 % on the
    SSD300 step in a world of one rank with a private stream, gpurun r05y)."""

    def __init__(self, group=None, device=None, stream=None):
        from . import _lib
        self._lib = _lib
        lib = _lib.load()
        self.device = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
        on = dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        ident = ctypes.create_string_buffer(COMM_ID_BYTES)
        if self.rank == 0:
            _lib.check(lib.odtk_comm_unique_id(ident))
        if self.world > 1:
            box = [ident.raw]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            ident = ctypes.create_string_buffer(box[0], COMM_ID_BYTES)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.odtk_comm_init(ident, self.rank, self.world, ctypes.byref(handle)))
        self.stream = stream
        self.handle = handle
        rk, wd = ctypes.c_int(-1), ctypes.c_int(-1)
        _lib.check(lib.odtk_comm_info(self.handle, ctypes.byref(rk), ctypes.byref(wd)))
        assert (rk.value, wd.value) == (self.rank, self.world)

    def all_reduce(self, buf: torch.Tensor):
        assert buf.is_cuda and buf.is_contiguous() and buf.dtype in (torch.float32, torch.bfloat16), (buf.device, buf.dtype)
        cur = torch.cuda.current_stream()
        st = self.stream if self.stream is not None else cur
        if st != cur:
            st.wait_stream(cur)
        dt = self._lib.F32 if buf.dtype == torch.float32 else self._lib.BF16
        ptr = ctypes.c_void_p(buf.data_ptr())          # (a slice of the model's persistent gradient / staging buffer: nothing for the allocator to track)
        self._lib.check(self._lib.load().odtk_comm_allreduce(self.handle, ptr, ptr, buf.numel(), dt, ctypes.c_void_p(st.cuda_stream)))
        ev = torch.cuda.Event()
        ev.record(st)
        return _Done(ev)

    def close(self):
        if self.handle is not None:
            torch.cuda.synchronize(self.device)
            self._lib.check(self._lib.load().odtk_comm_destroy(self.handle))
            self.handle = None


class BucketAllReducer:
    """Device-agnostic core: flat buffer + ordered segment table -> bucketed async all-reduce."""

    def __init__(self, flat: torch.Tensor, segments, group=None, bucket_bytes=25 << 20, comm_dtype='f32', force_collectives=False, cast=None,
                 collective=None):
        """segments: list of (name, start, end) in ascending offset order covering `flat`.
        comm_dtype 'bf16': a bucket travels as a bf16 copy (half the bytes on the xGMI links: 52 instead of 105 MB per SSD300 step) -- cast
        (`cast` = (narrow, widen) launches, odtk.ops.cast_from_f32 / cast_to_f32 on the GPU), summed by the collective in bf16, widened back
        into the f32 buffer; the sum of W bf16 values carries ~log2(W) fewer good bits than the f32 path, the optimizer still runs in f32.
        force_collectives: issue the collectives even in a world of ONE rank (the RCCL code path exercised on a single GPU).
        collective: None = torch.distributed's all_reduce; an OdtkCollective = the C-ABI's (odtk_comm_allreduce)."""
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.collective = collective
        self.force = bool(force_collectives) and (dist.is_initialized() or collective is not None)
        self.comm_dtype = comm_dtype
        assert comm_dtype in ('f32', 'bf16')
        self.stage = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device) if comm_dtype == 'bf16' else None
        self._narrow, self._widen = cast if cast is not None else (lambda x, out: out.copy_(x), lambda x, out: out.copy_(x))
        self._staged = []
        self.segments = list(segments)
        esz = flat.element_size()
        # cut buckets from the END (first gradients to become ready)
        self.buckets = []          # (start, end, lowest_segment_index)
        end = self.segments[-1][2]
        cur_lo = len(self.segments)
        size = 0
        for i in reversed(range(len(self.segments))):
            _, s, e = self.segments[i]
            size += (e - s) * esz
            cur_lo = i
            if size >= bucket_bytes or i == 0:
                self.buckets.append((s, end, cur_lo))
                end = s
                size = 0
        self.index = {name: i for i, (name, _, _) in enumerate(self.segments)}
        self.enabled = True            # False: walk the buckets without launching collectives (bench.py: step time without comm)
        # optional callable -> context manager under which a bucket's narrowing cast and collective are issued (ssd300._comm_launch: a launch stream that
        # waits for every stream carrying gradient kernels, so that the model keeps its side streams under data parallel); None = the current stream
        self.launch_ctx = None
        self.begin_step()

    def begin_step(self):
        self.next_bucket = 0
        self.lowest_ready = len(self.segments)
        self.handles = []
        self.launch_log = []           # (start, end) of the buckets in the order this step closed them (tests: the order does not depend on the streams)
        self._staged = []

    def segment_ready(self, name):
        """Call when the gradient of segment `name` (and every later segment) is final."""
        self.lowest_ready = min(self.lowest_ready, self.index[name])
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][2] >= self.lowest_ready:
            s, e, _ = self.buckets[self.next_bucket]
            if (self.world > 1 or self.force) and self.enabled:
                with (self.launch_ctx() if self.launch_ctx is not None else contextlib.nullcontext()):
                    buf = self.flat[s:e]
                    if self.stage is not None:
                        buf = self.stage[s:e]
                        self._narrow(self.flat[s:e], buf)      # on the launching stream, which the collective's stream waits for
                        self._staged.append((s, e))
                    self.handles.append(self._all_reduce(buf))
            self.launch_log.append((s, e))
            self.next_bucket += 1

    def _all_reduce(self, buf):
        if self.collective is not None:
            return self.collective.all_reduce(buf)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def bucket_bytes(self):
        return [(e - s) * self.flat.element_size() for s, e, _ in self.buckets]

    def all_reduce_alone(self):
        """Every bucket's all-reduce back to back with nothing to overlap with.  TIMING ONLY (bench.py: the collective's own time): it sums
        whatever the gradient buffer holds and, with a narrowed communication dtype, leaves the result in the staging copy without widening it
        back -- never call it inside a training step."""
        assert not self.handles and not self._staged, 'all_reduce_alone() inside a step: it is a timing helper, not part of the data path'
        if self.world > 1 or self.force:
            src = self.flat
            if self.stage is not None:
                src = self.stage
                self._narrow(self.flat, self.stage)          # the same bytes and arithmetic as the real path (never uninitialised memory)
            hs = [self._all_reduce(src[s:e]) for s, e, _ in self.buckets]
            for h in hs:
                h.wait()

    def finish_step(self):
        """Flush whatever is left and make the reduced gradients visible to the current stream."""
        self.lowest_ready = 0
        self.segment_ready(self.segments[0][0])
        for h in self.handles:
            h.wait()
        self.handles = []
        for s, e in self._staged:                              # after wait(): ordered behind the collectives on the current stream
            self._widen(self.stage[s:e], self.flat[s:e])
        self._staged = []


class GradAllReducer:
    """Binds a BucketAllReducer to an SSD300 instance (layer-granular readiness)."""

    def __init__(self, model, group=None, bucket_mb=25, grad_dtype='f32', force_collectives=False, collective='torch'):
        self.model = model
        segs = []
        names = list(model.pinfo.keys())
        # one segment per layer: from its first parameter to the next layer's first parameter
        layer_start = {}
        for n in names:
            layer = n.split('.')[0]
            layer_start.setdefault(layer, model.pinfo[n][0])
        layers = list(layer_start.keys())
        for i, l in enumerate(layers):
            end = layer_start[layers[i + 1]] if i + 1 < len(layers) else model.nparam
            segs.append((l, layer_start[l], end))
        cast = None
        if grad_dtype == 'bf16' and model.G.is_cuda:
            from . import ops
            cast = (ops.cast_from_f32, ops.cast_to_f32)
        # collective: 'torch' | 'odtk' | an OdtkCollective the caller created earlier (bench.py does, BEFORE the model: see OdtkCollective)
        # NOTE (round-5 advisory): the C-ABI collective is enqueued on the model's `_side` stream -- the stream box matching and the filter refresh run on
        # under the forward pass -- unless the OdtkCollective was built with its own: the bucket sums of step t are therefore ordered BEFORE the box matching
        # of step t + 1 on that stream (they do not overlap each other; both overlap the main chain).  Give OdtkCollective(stream=...) a stream of its
        # own to lift that; measured neutral in a world of one rank (profiles/r05w).
        if isinstance(collective, OdtkCollective):
            coll = collective
            if coll.stream is None:
                coll.stream = getattr(model, '_side', None)
        else:
            assert collective in ('torch', 'odtk'), collective
            coll = OdtkCollective(group, model.G.device, getattr(model, '_side', None)) if collective == 'odtk' else None
        self.red = BucketAllReducer(model.G, segs, group, int(bucket_mb) << 20, grad_dtype, force_collectives, cast, coll)
        self.world = self.red.world

    def boundary_layers(self):
        """Layers whose readiness closes a bucket (the backward graph is cut behind them)."""
        return [self.red.segments[lo][0] for (_, _, lo) in self.red.buckets]

    def begin_step(self):
        self.red.begin_step()

    def layer_ready(self, layer):
        self.red.segment_ready(layer)

    def finish_step(self):
        self.red.finish_step()
