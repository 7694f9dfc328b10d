"""TEST INFRASTRUCTURE ONLY: the non-MFMA kernels of libodtk compiled with g++ against tests/hip_cpu/hip/hip_runtime.h (workgroups run one after the other, their
threads as fibers that switch at __syncthreads / wave shuffles / ballots) and bound to odtk.ops, so that the test bodies written for the GPU run on the CPU from
the SAME kernel source.  `with installed():` patches, for the entry points the CPU build exports (csrc/lhrcnn.hip, csrc/augment.hip), the pointer / stream helpers
of odtk.ops and its C-ABI call; odtk_nms_batched (csrc/boxes.hip: DPP and LDS-DMA code, not emulated) is replaced by the oracle's NMS with the kernel's operand
addressing."""
import contextlib
import ctypes as C
import os
import subprocess
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, 'object-detection-tensorflow_amd', 'csrc')
SOURCES = [os.path.join(CSRC, 'lhrcnn.hip'), os.path.join(CSRC, 'augment.hip'), os.path.join(HERE, 'hip_cpu', 'stubs.cpp')]
_LIB = None


def build():
    """g++ build of the kernel sources, cached next to the temp dir by the newest source time stamp"""
    global _LIB
    if _LIB is not None:
        return _LIB
    deps = SOURCES + [os.path.join(HERE, 'hip_cpu', 'hip', 'hip_runtime.h'), os.path.join(CSRC, 'common.h'), os.path.join(CSRC, 'augment_resize.h'),
                      os.path.join(ROOT, 'include', 'odtk.h')]
    stamp = int(max(os.path.getmtime(f) for f in deps))
    so = os.path.join(tempfile.gettempdir(), f'libodtk_cpu_{os.getuid()}_{stamp}.so')
    if not os.path.exists(so):
        tmp = so + f'.{os.getpid()}.tmp'
        subprocess.check_call(['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-x', 'c++', '-I', os.path.join(HERE, 'hip_cpu'), '-I', CSRC]
                              + SOURCES + ['-o', tmp])
        os.replace(tmp, so)
    _LIB = C.CDLL(so)
    return _LIB


def _nms_batched(boxes, box_stride, scores, score_bstride, score_estride, valid, valid_bstride, valid_estride, valid_value, n, B, max_out_dev, max_out_stride,
                 max_out_const, iou_thr, out_idx, cap, out_cnt):
    sys.path.insert(0, ROOT)
    from oracle import ssd300_ref as R
    bx, sc = boxes.reshape(-1), scores.reshape(-1)
    vd = valid.reshape(-1) if valid is not None else None
    idx = torch.arange(n)
    for b in range(B):
        bb = bx[b * box_stride: b * box_stride + 4 * n].reshape(n, 4)
        ss = sc[b * score_bstride + idx * score_estride]
        ok = torch.ones(n, dtype=torch.bool) if vd is None else (vd[b * valid_bstride + idx * valid_estride] == valid_value)
        rows = torch.nonzero(ok).flatten()
        k_max = int(max_out_dev.reshape(-1)[b * max_out_stride]) if max_out_dev is not None else int(max_out_const)
        sel = torch.from_numpy(R.nms(bb[rows].numpy(), ss[rows].numpy(), k_max, float(iou_thr)).astype('int64'))
        k = min(sel.numel(), cap)
        out_idx.view(B, cap)[b, :k] = rows[sel[:k]].to(out_idx.dtype)
        out_cnt.view(-1)[b] = k


@contextlib.contextmanager
def installed():
    import odtk  # noqa: F401
    from odtk import _lib, ops
    lib = build()
    names = [n for n in _lib.SIGNATURES if hasattr(lib, n)]
    for n in names:
        f = getattr(lib, n)
        f.restype, f.argtypes = _lib.SIGNATURES[n]

    def call(name, *args):
        if not hasattr(lib, name):
            raise RuntimeError(f'{name} is not part of the CPU-emulated build (only csrc/lhrcnn.hip and csrc/augment.hip are)')
        rc = getattr(lib, name)(*args)
        if rc != 0:
            raise _lib.OdtkError(f'libodtk (CPU emulation) error {rc}: {lib.odtk_last_error().decode()}')
    old = dict(call=ops.call, _p=ops._p, _stream=ops._stream, nms=ops.nms_batched, sync=torch.cuda.synchronize)
    ops.call = call
    ops._p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    ops._stream = lambda: None
    ops.nms_batched = _nms_batched
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        yield names
    finally:
        ops.call, ops._p, ops._stream, ops.nms_batched, torch.cuda.synchronize = old['call'], old['_p'], old['_stream'], old['nms'], old['sync']
