"""In-situ shadow of every launch of a model class (TEST INFRASTRUCTURE ONLY).

`Shadow` wraps the launching functions of odtk.ops.  While it is recording, every launch is executed twice: by libodtk on the engine's own
buffers, and by the plain-PyTorch f32 restatement of the same op (tests/mock_ops.py: rows x pitch operands, pad columns, accumulate flags,
in-place outputs) on CLONES of those buffers taken right before the launch.  The declared outputs of the launch are then compared.  Because
the reference computation of every launch starts from the engine's OWN stored inputs, rounding does not accumulate from layer to layer
and the bound is that of one launch (one bf16 store, or f32 accumulation order) -- a wrong tile, a 32-bit offset overflow, a mis-split K range
or a stale split-K partial at the shapes the model really runs fails by orders of magnitude.  This is the harness of
tests/test_gpu_ssd300_b32.py::test_bf16_engine_every_layer_in_situ_at_batch32 made independent of the model: it sees launches, not layers,
so it serves SSD300, YOLOv3, RetinaNet, FCOS, CenterNet ... at their BASELINE shapes alike.

Where the shadow runs: dense ops (convolutions, norms, pools, resizes, element-wise, optimizer) on the GPU in f32 through torch's NATIVE kernels
(im2col + GEMM: `torch.backends.cudnn.flags(enabled=False)` keeps MIOpen out, so nothing is tuned or compiled at run time); the box-side
losses on the CPU through the oracles (oracle/*_ref.py), exactly as tests/mock_ops.py calls them."""
import contextlib
import inspect
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
import mock_ops  # noqa: E402

# output parameters of every launch (names of tests/mock_ops.py's signatures); 'loss_parts:K' = only column(s) K of that tensor
OUTPUTS = {
    'conv2d_fwd': ['y'], 'conv2d_fwd_pool2x2': ['y', 'y_pool'], 'conv2d_dgrad': ['dx'], 'conv2d_wgrad': ['dw', 'dbias'],
    'conv2d_fwd_bits': ['y'], 'conv2d_dgrad_bits': ['dx'],
    'colsum': ['out'],
    'preprocess': ['x'], 'preprocess_norm': ['x'],
    'bn_fwd': ['mmean', 'mvar', 'save_mean', 'save_invstd', 'y'], 'bn_bwd': ['dz', 'dgamma', 'dbeta'],
    'gn_fwd': ['y', 'save'], 'gn_bwd': ['dx', 'dgamma', 'dbeta'],
    'add2d': ['y'], 'add_relu_fwd': ['y'], 'relu_bwd': ['dx'],
    'upsample2x_fwd': ['y'], 'upsample2x_bwd': ['dx'],
    'resize_bilinear_fwd': ['y'], 'resize_bilinear_bwd': ['dx'], 'resize_bilinear2_fwd': ['y'], 'resize_bilinear2_bwd': ['dx'],
    'copy_channels': ['dst'],
    'maxpool_fwd': ['y'], 'maxpool_bwd': ['dx'], 'maxpool_fwd_argmax': ['y'], 'maxpool_bwd_argmax': ['dx'], 'maxpool2x2_fwd_idx': ['y'], 'maxpool2x2_bwd_idx': ['dx'],
    'avgpool2x2_fwd': ['y'], 'avgpool2x2_bwd': ['dx'],
    'rows_to_f32': ['y'], 'rows_from_f32': ['x'], 'exp_rows_to_f32': ['y'], 'exp_rows_bwd': ['dx'],
    'l2norm_fwd': ['y'], 'l2norm_bwd': ['dx', 'dgamma'],
    'sgd_momentum': ['p', 'm', 'p_cast', 'l2_partial:sum'], 'adam': ['p', 'm', 'v', 'p_cast', 'l2_partial:sum'],
    'sum_f32': ['out'], 'zero': ['x'], 'loss_total': ['sum_a', 'sum_b', 'total'], 'cast_from_f32': ['out'], 'cast_to_f32': ['out'],
    'ssd_loss': ['loss_parts:3', 'dpred'], 'yolov3_loss': ['loss_parts:4', 'd_preds'], 'yolov2_loss': ['loss_parts:4', 'd_pred'],
    'retina_loss': ['loss_parts:0,1', 'dconf', 'dbox'], 'fcos_loss': ['loss', 'd_conf', 'd_reg', 'd_center'],
    'centernet_loss': ['loss_parts:3', 'd_keypoints', 'd_offset', 'd_size'],
    'refinedet_loss': ['loss_parts:6', 'd_arm_loc', 'd_arm_conf', 'd_odm_loc', 'd_odm_conf'],
    'depthwise_conv': ['y'], 'depthwise_wgrad': ['dfilt'], 'crop_and_resize_fwd': ['out'], 'crop_and_resize_bwd': ['d_feat'],
    'lhrcnn_rpn_loss': ['d_conf', 'd_bbox'], 'lhrcnn_rcnn_loss': ['d_logits', 'd_pbbox'],
}
# launches whose restatement calls the CPU oracles
ON_CPU = {'ssd_loss', 'yolov3_loss', 'yolov2_loss', 'retina_loss', 'fcos_loss', 'centernet_loss', 'refinedet_loss', 'lhrcnn_rpn_loss'}
# launches that are not shadowed: the mocked box-side front ends do nothing (the mocked loss matches / mines by itself through the oracle -- the
# real kernels' indices are compared bit for bit by the kernel-level tests), workspaces, scratch selection, constant tables
PASS = {'conv2d_fwd_pool2x2_fused', 'conv2d_x3_supported', 'conv2d_relu_bits_supported', 'ssd_match', 'softmax_ce_const', 'retina_match', 'nms_batched', 'scratch_slot', 'ssd_priors', 'retina_anchors', 'gn_workspace', 'fcos_workspace',
        'yolov3_workspace', 'retina_match_workspace', 'centernet_workspace', 'yolov3_decode_candidates', 'fcos_decode_candidates', 'retina_decode',
        'refinedet_decode', 'centernet_decode', 'yolov2_decode_candidates', 'lhrcnn_match', 'lhrcnn_rpn_decode', 'lhrcnn_gather_rois', 'lhrcnn_rcnn_decode'}
WHOLE_STORAGE_MAX = 1 << 30
BN_MOM_ = 0.99


# Reductions whose terms cancel (the bias gradient of a convolution in front of a batch norm is sum(dz) = 0 in exact arithmetic; dgamma / dbeta of a
# normalisation likewise sum signed terms): both sides return round-off, and an error relative to the RESULT says nothing.  For these outputs the
# error is taken relative to the sum of the ABSOLUTE terms (what the forward error bound of a summation is stated against).
def _bn_terms(a):
    d = mock_ops._read_rows(a['dy'], a['M'], a['C_'], a['ldy'], a['rows_per_img'], a['y_img_stride'])
    if a['relu']:
        yv = mock_ops._read_rows(a['y'], a['M'], a['C_'], a['ldy'], a['rows_per_img'], a['y_img_stride'])
        d = d * (yv > 0) if a['relu'] == 1 else torch.where(yv > 0, d, 0.1 * d)
    xh = (a['z'][:a['M'], :a['C_']].float() - a['save_mean']) * a['save_invstd']
    return d, xh


def _gn_terms(a):
    d = a['dy'][:, :a['C_']].float()
    if a['relu']:
        d = d * (a['y'][:, :a['C_']].float() > 0)
    xh = mock_ops._gn_xhat(a['x'], a['N'], a['HW'], a['C_'], a['groups'], a['save'][:, :, 0], a['save'][:, :, 1]).reshape(a['N'] * a['HW'], a['C_'])
    return d, xh


COND = {
    # a batch mean is a cancelling sum too (PFPNetR's 85-channel branches: |mean| ~ 1e-6 of the spread): against the mean of |z|
    ('bn_fwd', 'save_mean'): lambda a: a['z'][:a['M'], :a['C_']].float().abs().mean(0),
    ('bn_fwd', 'mmean'): lambda a: BN_MOM_ * a['mmean'].float().abs() + (1 - BN_MOM_) * a['z'][:a['M'], :a['C_']].float().abs().mean(0),
    ('conv2d_wgrad', 'dbias'): lambda a: a['dy'][:, :a['d'].K].float().abs().sum(0),
    ('colsum', 'out'): lambda a: a['dy'][:a['M'], :a['C_']].float().abs().sum(0),
    ('bn_bwd', 'dbeta'): lambda a: _bn_terms(a)[0].abs().sum(0),
    ('bn_bwd', 'dgamma'): lambda a: (lambda d, xh: (d * xh).abs().sum(0))(*_bn_terms(a)),
    ('l2norm_bwd', 'dgamma'): lambda a: (lambda v, d: (d * v / torch.sqrt(torch.clamp((v * v).sum(1, keepdim=True), min=1e-12))).abs().sum().view(1))(
        a['x'][:a['M'], :a['C_']].float(), a['dy'][:a['M'], :a['C_']].float()),
    ('gn_bwd', 'dbeta'): lambda a: _gn_terms(a)[0].abs().sum(0),
    ('gn_bwd', 'dgamma'): lambda a: (lambda d, xh: (d * xh).abs().sum(0))(*_gn_terms(a)),
}


def _map(obj, f):
    if isinstance(obj, torch.Tensor):
        return f(obj)
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(o, f) for o in obj)
    if isinstance(obj, dict):                                # LHRCNN hands its anchors and its workspace over as dicts of tensors
        return {k: _map(v, f) for k, v in obj.items()}
    return obj


def _flat(obj, out):
    if isinstance(obj, torch.Tensor):
        out.append(obj)
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            _flat(o, out)
    return out


class Shadow:
    def __init__(self):
        self.recording = False
        self.records = []            # dicts: op, seq, out, rel (device scalar), numel, dtype, shape
        self.seq = 0
        self.unknown = set()
        self.native_outside = {}     # C-ABI entry points called while recording but NOT from inside a shadowed launch (coverage report)
        self.native_inside = {}
        self._depth = 0
        self._old = {}
        self._sig = {}

    # ------------------------------------------------------------------ installation
    @contextlib.contextmanager
    def installed(self):
        """patch odtk.ops (before the model is constructed: FilterPrepareBatch objects are created then)"""
        from odtk import ops
        was = torch.backends.cudnn.enabled
        names = [n for n, v in vars(mock_ops).items() if callable(v) and not n.startswith('_') and n not in ('installed', 'contextlib') and hasattr(ops, n)]
        try:
            for n in names:
                real = getattr(ops, n)
                self._old[n] = real
                if n == 'FilterPrepareBatch':
                    setattr(ops, n, self._wrap_filter_prepare(real))
                elif n not in PASS:
                    self._sig[n] = inspect.signature(getattr(mock_ops, n))
                    setattr(ops, n, self._wrap(n, real, getattr(mock_ops, n)))
            was, torch.backends.cudnn.enabled = torch.backends.cudnn.enabled, False      # MIOpen out: the restatement runs torch's native kernels
            native = ops.call

            def counted(fn, *a):
                if self.recording:
                    d = self.native_inside if self._depth else self.native_outside
                    d[fn] = d.get(fn, 0) + 1
                return native(fn, *a)
            ops.call = counted
            try:
                yield self
            finally:
                ops.call = native
        finally:
            torch.backends.cudnn.enabled = was
            for n, v in self._old.items():
                setattr(ops, n, v)
            self._old = {}

    # ------------------------------------------------------------------ clones
    @staticmethod
    def _clone_dev(t, memo):
        """device clone that keeps the storage geometry (the kernels get raw pointers and address past the view they were handed, e.g. a batch-norm
        that writes a head's rows into the middle of the [N][A][25] prediction tensor): the whole storage is cloned once per launch and re-viewed"""
        st = t.untyped_storage()
        if st.nbytes() > WHOLE_STORAGE_MAX:
            return t.detach().clone()
        key = st.data_ptr()
        if key not in memo:
            memo[key] = torch.empty(0, dtype=torch.uint8, device=t.device).set_(st).clone()
        base = memo[key]
        flat = torch.empty(0, dtype=t.dtype, device=t.device).set_(base.untyped_storage())
        return flat.as_strided(t.shape, t.stride(), t.storage_offset())

    # ------------------------------------------------------------------ comparison
    def _compare(self, op, pname, real, mock, snap, col=None, scale=None):
        """relative error (Frobenius) of the engine's output against the restatement's, over the elements the restatement wrote; `scale`: the
        sum of absolute terms of a cancelling reduction (COND), which then replaces the restatement's own norm as the denominator"""
        if real is None or mock is None:
            return
        for r, m, s in zip(_flat(real, []), _flat(mock, []), _flat(snap, [])):
            r = r.detach(); m = m.detach().to(r.device); s = s.detach().to(r.device)
            if col == 'sum':
                r, m, s = r.double().sum().view(1), m.double().sum().view(1), s.double().sum().view(1) - 1.0
            elif col is not None:
                idx = [int(c) for c in col.split(',')]
                r, m, s = r[:, idx], m[:, idx], s[:, idx]
            if r.dtype in (torch.uint8, torch.int32, torch.int64, torch.bool):
                bad = (r != m).sum()
                self.records.append(dict(op=op, seq=self.seq, out=pname, rel=bad.float(), numel=r.numel(), dtype=str(r.dtype), exact=True, shape=tuple(r.shape)))
                continue
            rf, mf, sf = r.double(), m.double(), s.double()
            wrote = (mf != sf) if col != 'sum' else torch.ones_like(mf, dtype=torch.bool)
            diff = torch.where(wrote, rf - mf, torch.zeros_like(rf))
            num = torch.sqrt((diff * diff).sum())
            den = torch.sqrt(torch.where(wrote, mf * mf, torch.zeros_like(mf)).sum())
            if scale is not None:
                sc = scale.detach().double().reshape(-1)[: mf.numel()]
                den = torch.sqrt((sc * sc).sum())
            stray = (~wrote & (rf != sf)).sum()           # elements the engine changed and the restatement did not
            self.records.append(dict(op=op, seq=self.seq, out=pname, rel=(num / (den + 1e-30)).float(), den=den.float(), wrote=wrote.sum(), stray=stray,
                                     numel=r.numel(), dtype=str(r.dtype), exact=False, shape=tuple(r.shape)))

    # ------------------------------------------------------------------ wrappers
    def _wrap(self, name, real, mock):
        sig = self._sig[name]

        def f(*args, **kw):
            if not self.recording:
                return real(*args, **kw)
            if name not in OUTPUTS:
                self.unknown.add(name)
                return real(*args, **kw)
            bound = sig.bind(*args, **kw)
            bound.apply_defaults()
            on_cpu = name in ON_CPU
            memo = {}
            conv = (lambda t: t.detach().cpu().clone()) if on_cpu else (lambda t: self._clone_dev(t, memo))
            margs = {k: _map(v, conv) for k, v in bound.arguments.items()}
            outs = [o.split(':') for o in OUTPUTS[name]]
            snap = {o[0]: _map(margs.get(o[0]), lambda t: t.clone()) for o in outs}
            self._depth += 1
            try:
                r = real(*args, **kw)
            finally:
                self._depth -= 1
            if name == 'maxpool2x2_bwd_idx':              # the restatement keeps torch's arg-max of the forward launch, keyed by the index buffer
                mock_ops._POOL_ARGMAX[margs['idx'].data_ptr()] = mock_ops._POOL_ARGMAX[bound.arguments['idx'].data_ptr()]
            if name == 'maxpool_bwd_argmax':
                mock_ops._POOL_ARGMAX[margs['arg'].data_ptr()] = mock_ops._POOL_ARGMAX[bound.arguments['arg'].data_ptr()]
            mock(**margs)
            if name in ('maxpool2x2_fwd_idx', 'conv2d_fwd_pool2x2') and margs.get('idx') is not None:
                mock_ops._POOL_ARGMAX[bound.arguments['idx'].data_ptr()] = mock_ops._POOL_ARGMAX.pop(margs['idx'].data_ptr())
            if name == 'maxpool_fwd_argmax':
                mock_ops._POOL_ARGMAX[bound.arguments['arg'].data_ptr()] = mock_ops._POOL_ARGMAX.pop(margs['arg'].data_ptr())
            for o in outs:
                cond = COND.get((name, o[0]))
                scale = cond(margs) if (cond is not None and margs.get(o[0]) is not None) else None
                self._compare(name, o[0], bound.arguments.get(o[0]), margs.get(o[0]), snap.get(o[0]), o[1] if len(o) > 1 else None, scale)
            self.seq += 1
            return r
        f.__name__ = name
        return f

    def _wrap_filter_prepare(self, real_cls):
        shadow = self

        class FilterPrepareBatch(real_cls):
            def run(self):
                if not shadow.recording:
                    return super().run()
                memo = {}
                keep = getattr(self, 'keep', None) or self.entries       # (the CPU stand-in of the class calls them `entries`)
                entries = [(shadow._clone_dev(w, memo), shadow._clone_dev(wt, memo), K, R, S, C_, Kp) for (w, wt, K, R, S, C_, Kp) in keep]
                snaps = [e[1].clone() for e in entries]
                shadow._depth += 1
                try:
                    super().run()
                finally:
                    shadow._depth -= 1
                mock_ops.FilterPrepareBatch(entries, getattr(self, 'dtype', None), None).run()
                for i, (e, sn) in enumerate(zip(entries, snaps)):
                    shadow._compare('filter_prepare', f'wt[{i}]', keep[i][1], e[1], sn)
                shadow.seq += 1
        return FilterPrepareBatch

    # ------------------------------------------------------------------ report
    def summary(self):
        """(rows sorted by error, launches shadowed) -- one host sync here, none per launch"""
        rows = []
        for rec in self.records:
            row = dict(rec)
            row['rel'] = float(rec['rel'])
            for k in ('den', 'wrote', 'stray'):
                if k in row:
                    row[k] = float(row[k])
            rows.append(row)
        rows.sort(key=lambda r_: -r_['rel'])
        return rows, self.seq

    def check(self, tol_of, verbose=True, label=''):
        """tol_of(row) -> bound of that output; raises AssertionError listing every violation"""
        rows, n = self.summary()
        bad = []
        for r_ in rows:
            if r_['exact']:
                if r_['rel'] != 0:
                    bad.append((r_['op'], r_['seq'], r_['out'], 'mismatching integer elements', r_['rel'], r_['shape']))
            elif not (r_['rel'] <= tol_of(r_)):            # NaN fails
                bad.append((r_['op'], r_['seq'], r_['out'], r_['rel'], tol_of(r_), r_['shape']))
        if verbose:
            per_op = {}
            for r_ in rows:
                a = per_op.setdefault(r_['op'] + ':' + r_['out'].split('[')[0], [0, 0.0, 0.0])
                a[0] += 1; a[1] = max(a[1], r_['rel']); a[2] += r_.get('stray', 0.0)
            print(f'[in-situ {label}] {n} launches shadowed, {len(rows)} outputs compared; worst relative error per launch kind:')
            for k, (cnt, worst, stray) in sorted(per_op.items(), key=lambda kv: -kv[1][1]):
                print(f'    {k:38s} x{cnt:<4d} worst {worst:.3e}' + (f'   stray elements {int(stray)}' if stray else ''))
            if self.unknown:
                print('    NOT shadowed (no output table entry):', sorted(self.unknown))
            print('    C-ABI calls inside shadowed launches:', sum(self.native_inside.values()), '; outside (box-side front ends, scratch selection):',
                  dict(sorted(self.native_outside.items())))
        assert not self.unknown, f'launches without an output table entry: {sorted(self.unknown)}'
        assert n > 0 and rows, 'nothing was shadowed'
        assert not bad, f'{len(bad)} outputs out of bound, worst first: {bad[:12]}'
        return rows


def default_tol(engine_dtype):
    """bounds of ONE launch, set from what the BASELINE-shape runs measure (profiles/r03_insitu_configs.md) with a 5-10x margin -- a wrong tile or
    a stale partial is O(1).  Both sides store through the same rounding, so a bf16 output differs only where the f32 accumulation order moves a value
    across a rounding boundary (measured 4-7e-5); a bf16 buffer that is ACCUMULATED into is rounded twice by the engine and once by the restatement
    (measured 2.8e-3 = one bf16 rounding of the first addend).  f32 engine: accumulation order only, f32 atomics over up to 2.5 M pixels included
    (measured <= 1e-5).  Box-side losses and their gradients against the oracles on the engine's own logits: <= 1e-6 measured."""
    # (maxpool2x2_bwd_idx: behind a FUSED conv + pool the restatement pools its own f32 convolution -- a window whose two largest values round to neighbouring
    #  bf16 numbers in one accumulation order and to the same one in the other routes its gradient to another position: 7 of 11.5 M windows for conv3_3 + pool3
    #  at batch 32 = 2.2e-3 in the Frobenius measure; a wrong routing table would be O(1))
    twice = ('conv2d_dgrad', 'conv2d_dgrad_bits', 'maxpool2x2_bwd_idx', 'add2d', 'relu_bwd', 'upsample2x_bwd', 'resize_bilinear_fwd', 'resize_bilinear_bwd', 'copy_channels', 'l2norm_bwd', 'gn_bwd',
             'bn_bwd', 'maxpool_bwd', 'resize_bilinear2_bwd', 'resize_bilinear2_fwd')

    def tol(row):
        if row['op'] in ON_CPU:
            return 1e-4
        if engine_dtype in ('f32', 'f32x3'):            # (x3: products carry 2^-17; in the Frobenius measure of one launch 4-8e-6 measured)
            return 5e-5
        if row['dtype'] == 'torch.bfloat16':
            return 6e-3 if row['op'] in twice else 4e-4
        return 2e-4
    return tol
