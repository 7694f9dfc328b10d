"""Eager TensorFlow-1.x API shim on torch-CPU -- TEST INFRASTRUCTURE ONLY.

Purpose: execute the reference's OWN Python (`/root/reference/SSD300.py`, loaded from disk at
fixture-generation time, never copied into this repo) so that the oracle restatement in
`oracle/ssd300_ref.py` can be pinned against what the reference code actually computes
(`tests/golden/make_golden.py`).

What this shim is and is not:
  * it implements only the `tf.*` symbols SSD300.py touches, eagerly, on torch CPU float32;
  * graph semantics are emulated by RE-TRACING: `Session.run(fetches, feed_dict)` re-executes
    `_define_inputs()` + `_build_graph()` of the model with the fed placeholder values and returns
    the freshly computed attributes that correspond to the requested fetches;
  * the numerical semantics of each TF kernel (SAME padding, fused batch-norm, NMSv3, arg-max tie
    rules, MomentumOptimizer ...) are restated from the TF 1.13 sources from memory
    (SURVEY.md Appendix B): that layer remains "parity unpinned".
"""
from __future__ import annotations

import math
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

float32, int32, int64, bool_ = torch.float32, torch.int32, torch.int64, torch.bool


class TFTensor(torch.Tensor):
    """torch.Tensor with the few tf.Tensor methods the reference calls."""

    def set_shape(self, shape):            # noqa: D401
        return None

    def get_shape(self):
        return list(self.shape)

    # tf.Tensor arithmetic converts a Python list operand (LH_RCNN.py:146: `pos_proposal / norm_factor`)
    def __truediv__(self, o):
        return torch.Tensor.__truediv__(self, _t(o, self.dtype) if isinstance(o, (list, tuple)) else o)

    def __mul__(self, o):
        return torch.Tensor.__mul__(self, _t(o, self.dtype) if isinstance(o, (list, tuple)) else o)


def _t(x, dtype=None):
    if isinstance(x, torch.Tensor):
        return x if dtype is None else x.to(dtype)
    return torch.as_tensor(np.asarray(x), dtype=dtype) if dtype is not None else torch.as_tensor(np.asarray(x))


def wrap(t):
    return torch.Tensor._make_subclass(TFTensor, t) if not isinstance(t, TFTensor) else t


# ---------------------------------------------------------------------------- state
class _State:
    def __init__(self):
        self.reset()

    def reset(self):
        self.variables = {}            # full name -> torch tensor (requires_grad for trainables)
        self.trainable = []            # names in creation order
        self.scope = []
        self.feeds = {}                # placeholder name -> value
        self.ph_count = 0
        self.apply_updates = False     # True while Session.run executes a fetch list containing train_op
        self.pending = []              # deferred variable updates (optimizer, BN moving stats)
        self.model = None
        self.momentum = {}
        self.uid = 0
        self.optimizer_scope = None    # variable scope open when optimizer.minimize() ran (prefix of slot / accumulator variable names)


S = _State()


def reset():
    S.reset()


# ---------------------------------------------------------------------------- basic ops
def convert_to_tensor(v, dtype=None):
    if isinstance(v, (list, tuple)) and any(isinstance(e, torch.Tensor) for e in v):
        return torch.stack([_t(e, dtype) for e in v])
    return _t(v, dtype)


def constant(v, dtype=None):
    if dtype is None:
        arr = np.asarray(v)
        dtype = int32 if arr.dtype.kind in 'iu' else float32
    return _t(v, dtype)


def cast(x, dtype):
    return _t(x).to(dtype)


def shape(x):
    return list(x.shape)


def reshape(x, shp):
    shp = [int(s) for s in (shp.tolist() if isinstance(shp, torch.Tensor) else shp)]
    return x.reshape(shp)


def tile(x, multiples):
    return x.repeat(*[int(m) for m in multiples])


def concat(values, axis):
    return torch.cat([_t(v) for v in values], dim=axis)


def transpose(x, perm=None):
    return x.permute(*(perm if perm is not None else list(range(x.dim()))[::-1]))


def squeeze(x, axis=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


def range_(start, limit=None, delta=1, dtype=None):
    if limit is None:
        start, limit = 0, start
    f = lambda v: v.item() if isinstance(v, torch.Tensor) else v
    if dtype is None:
        dtype = float32 if any(isinstance(f(v), float) for v in (start, limit, delta)) else int32
    return torch.arange(f(start), f(limit), f(delta), dtype=dtype)


def maximum(a, b):
    return torch.maximum(_t(a, float32) if not isinstance(a, torch.Tensor) else a,
                         torch.as_tensor(b, dtype=a.dtype) if not isinstance(b, torch.Tensor) else b)


def minimum(a, b):
    return torch.minimum(a, torch.as_tensor(b, dtype=a.dtype) if not isinstance(b, torch.Tensor) else b)


def reduce_prod(x, axis=None):
    return x.prod() if axis is None else x.prod(dim=axis)


def reduce_sum(x, axis=None):
    return x.sum() if axis is None else x.sum(dim=axis)


def reduce_mean(x, axis=None):
    if isinstance(x, (list, tuple)):                          # tf.reduce_mean packs a list of scalars (YOLOv3.py:311)
        x = torch.stack([_t(v) for v in x])
    return x.mean() if axis is None else x.mean(dim=axis)      # mean of empty -> nan, as TF


def reduce_max(x, axis=None, keepdims=False):
    return x.max() if axis is None else x.max(dim=axis, keepdim=keepdims).values


def reduce_min(x, axis=None, keepdims=False):
    if isinstance(x, (list, tuple)):          # tf.reduce_min([a, b, c]) packs the list first
        x = torch.stack([_t(v) for v in x])
    return x.min() if axis is None else x.min(dim=axis, keepdim=keepdims).values


def floor(x):
    return torch.floor(x)


# ---- scripted randomness: tf.random_uniform / tf.random.uniform pop values pushed by the test (TF's RNG cannot be matched)
RANDOM_QUEUE = []


def random_uniform(shape, minval=0., maxval=1., dtype=None):
    n = int(np.prod(shape)) if len(shape) else 0
    if n:
        return torch.tensor([RANDOM_QUEUE.pop(0) for _ in range(n)], dtype=dtype or float32).reshape(list(shape))
    return torch.tensor(RANDOM_QUEUE.pop(0), dtype=dtype or float32)


def cos(x):
    return torch.cos(_t(x))


def sin(x):
    return torch.sin(_t(x))


def pad(x, paddings, mode='CONSTANT', constant_values=0.):
    flat = []
    for lo, hi in reversed([(int(a), int(b)) for a, b in paddings]):
        flat += [lo, hi]
    return F.pad(x, flat, value=float(constant_values))


def slice_(x, begin, size):
    idx = tuple(slice(int(b), int(b) + int(s)) for b, s in zip(begin, size))
    return x[idx]


def reverse(x, axis):
    return torch.flip(x, [int(a) for a in axis])


def one_hot(indices, depth):
    return F.one_hot(_t(indices).long(), int(depth)).to(float32)


def sqrt(x):
    return torch.sqrt(x)


def square(x):
    return x * x


def sigmoid(x):
    return torch.sigmoid(x)


def log_sigmoid(x):
    return torch.nn.functional.logsigmoid(x)


def equal(a, b):
    return _t(a) == b


def greater(a, b):
    return _t(a) > b


def zeros(shp, dtype=None):
    return torch.zeros([int(s) for s in shp], dtype=dtype or float32)


def meshgrid(a, b):
    """tf.meshgrid(x, y) (indexing='xy'): two [len(y), len(x)] tensors."""
    return [a.view(1, -1).expand(b.shape[0], -1).clone(), b.view(-1, 1).expand(-1, a.shape[0]).clone()]


class _KerasBackend:
    @staticmethod
    def binary_crossentropy(target, output, from_logits=False):
        # tf.keras.backend.binary_crossentropy(from_logits=True) = nn.sigmoid_cross_entropy_with_logits:
        # max(x, 0) - x * z + log(1 + exp(-|x|))
        assert from_logits
        x, z = output, target
        return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x)))


keras = types.SimpleNamespace(backend=_KerasBackend())


def _first_arg(x, axis, largest):
    # tf.argmax / tf.argmin: first occurrence on ties, int64
    x = x.detach()
    ext = x.max(dim=axis, keepdim=True).values if largest else x.min(dim=axis, keepdim=True).values
    hit = (x == ext)
    n = x.shape[axis]
    idx = torch.arange(n).view([-1 if d == (axis % x.dim()) else 1 for d in range(x.dim())]).expand_as(x)
    return torch.where(hit, idx, torch.full_like(idx, n)).min(dim=axis).values.to(int64)


def argmax(x, axis=0, output_type=None):
    r = _first_arg(x, axis, True)
    return r if output_type is None else r.to(output_type)


def argmin(x, axis=0, output_type=None):
    r = _first_arg(x, axis, False)
    return r if output_type is None else r.to(output_type)


GATHER_OOB_ZERO = False     # tf.gather's documented device difference: the CPU kernel raises on an out-of-range index, the GPU kernel stores 0


def gather(params, indices):
    idx = _t(indices).long()
    if GATHER_OOB_ZERO and idx.numel() and params.dim() >= 1:
        n = params.shape[0]
        ok = (idx >= 0) & (idx < n)
        if not bool(ok.all()):
            if n == 0:
                return torch.zeros(tuple(idx.shape) + tuple(params.shape[1:]), dtype=params.dtype)
            out = params[idx.clamp(0, n - 1)]
            return torch.where(ok.view(tuple(idx.shape) + (1,) * (params.dim() - 1)), out, torch.zeros_like(out))
    return params[idx]


def boolean_mask(x, mask, axis=None):
    m = _t(mask).bool()
    if axis is None or axis == 0:
        return x[m]
    return x[(slice(None),) * axis + (m,)]


def unique(x):
    vals, first = [], {}
    for i, v in enumerate(x.tolist()):
        if v not in first:
            first[v] = len(vals)
            vals.append(v)
    return torch.tensor(vals, dtype=x.dtype), torch.tensor([first[v] for v in x.tolist()], dtype=int32)


def gather_nd(params, indices):
    idx = _t(indices).long()
    return params[tuple(idx[..., i] for i in range(idx.shape[-1]))]


def clip_by_value(x, lo, hi):
    return torch.clamp(x, lo, hi)


def pow(x, y):                              # noqa: A001
    return torch.pow(_t(x), y)


def expand_dims(x, axis):
    return _t(x).unsqueeze(axis)


def zeros_like(x, dtype=None):
    return torch.zeros_like(x, dtype=dtype)


def ones_like(x, dtype=None):
    return torch.ones_like(_t(x), dtype=dtype)


def less(a, b):
    return _t(a) < b


def greater_equal(a, b):
    return _t(a) >= b


def add(a, b):
    return a + b


def add_n(xs):
    out = xs[0]
    for v in xs[1:]:
        out = out + v
    return out


def exp(x):
    return torch.exp(x)


def log(x):
    return torch.log(x)


def abs_(x):
    return torch.abs(x)


def where(c, a, b):
    return torch.where(c, a, b)


TRACE_DEAD_COND_BRANCHES = False


def cond(pred, true_fn, false_fn):
    """tf.cond: the taken branch's value.  TensorFlow BUILDS both branches (cond_v1 traces true_fn, then false_fn), so variables a
    branch creates exist -- and are trainable, regularised, saved -- even when the branch never runs (CenterNet._basic_block's
    1x1 shortcut conv, CenterNet.py:378-389).  With TRACE_DEAD_COND_BRANCHES the untaken branch is therefore executed for its
    variable creation only: its value and its pending updates (batch-norm moving statistics: ops of a dead branch do not run) are
    dropped.  Off by default: a dead branch must not consume scripted random draws (image_augmentor)."""
    take_true = bool(pred)
    if not TRACE_DEAD_COND_BRANCHES:
        return true_fn() if take_true else false_fn()
    out = None
    for is_true, fn in ((True, true_fn), (False, false_fn)):
        if is_true == take_true:
            out = fn()
        else:
            n = len(S.pending)
            try:
                with torch.no_grad():
                    fn()                                 # eager arithmetic of a branch that would not run may fail (empty tensors):
            except Exception:                            # noqa: BLE001 -- only its variable creation matters, graph building has no values
                pass
            del S.pending[n:]
    return out


def case(pred_fn_pairs, default=None, exclusive=False):
    """tf.case(exclusive=False): the value of the first pair whose predicate holds, else default().  NOTE (LH_RCNN.py:194-201): the
    reference's branch functions return operations / tensors that were built OUTSIDE the case (`lambda: train_rpn_op`); in a TF-1.x
    graph such an op is not gated by the predicate -- cond_v1 only adds a control edge from the branch's pivot identity to it -- so every
    one of them runs on every step whichever branch is selected.  The shim is eager: whatever was built before the case has already
    registered its updates, which is that behaviour."""
    for pred, fn in pred_fn_pairs:
        if bool(pred):
            return fn()
    return default()


def while_loop(cond_fn, body, loop_vars):
    vars_ = tuple(loop_vars)
    while bool(cond_fn(*vars_)):
        vars_ = tuple(body(*vars_))
    return vars_


def group(*args, **kw):
    return ('group', args)


# ---------------------------------------------------------------------------- variables / scopes
AUTO_REUSE = 'auto_reuse'          # tf.AUTO_REUSE: accepted by variable_scope, variables are looked up by name anyway


class variable_scope:
    """tf.variable_scope with TF 1.x's scope-count bookkeeping (variable_scope.py, _VariableScopeStore): entering a scope
    counts its full name (open_variable_scope); LEAVING it resets the counts of all of its sub-scopes
    (close_variable_subscopes) -- so default layer names ('conv2d', 'batch_normalization', 'GroupNorm') are numbered per
    ENCLOSING scope, and a scope that is entered again (FCOS.py:351,358 with reuse=tf.AUTO_REUSE) hands out the same default
    names again: its layers then find their variables by name, i.e. the five pyramid levels SHARE one set of head weights.
    (Without reuse TensorFlow raises 'Variable ... already exists' there; the shim looks variables up by name either way.)"""

    def __init__(self, name, *a, default_name=None, **k):
        self.name = name if name is not None else _unique_scope_name(default_name)

    def __enter__(self):
        S.scope.append(self.name)
        full = '/'.join(S.scope)
        _SCOPE_COUNT[full] = _SCOPE_COUNT.get(full, 0) + 1
        return self

    def __exit__(self, *exc):
        full = '/'.join(S.scope)
        for k in list(_SCOPE_COUNT):
            if k.startswith(full + '/'):
                _SCOPE_COUNT[k] = 0
        S.scope.pop()


_SCOPE_COUNT = {}


def _unique_scope_name(prefix):
    """variable_scope(None, default_name=prefix): _get_unique_variable_scope -- prefix, prefix_1, ... among the scopes opened
    (and not yet reset) under the CURRENT scope.  This is how tf.layers names a layer created without `name=`
    (Layer._set_scope: variable_scope(None, default_name=self._base_name)) and how contrib's group_norm names its scope."""
    cur = '/'.join(S.scope)
    name = (cur + '/' + prefix) if cur else prefix
    if _SCOPE_COUNT.get(name, 0) == 0:
        return prefix
    idx = 1
    while _SCOPE_COUNT.get(f'{name}_{idx}', 0) > 0:
        idx += 1
    return f'{prefix}_{idx}'


def _full(name):
    return '/'.join(S.scope + [name])


class _Init:
    def __init__(self, fn):
        self.fn = fn

    def __call__(self, shp):
        return self.fn(shp)


def constant_initializer(v):
    return _Init(lambda shp: torch.full([int(s) for s in shp], float(v)))


def get_variable(name, shape=None, initializer=None, trainable=True, dtype=None):
    full = _full(name)
    if full not in S.variables:
        if isinstance(initializer, _Init):
            val = initializer(shape)
        elif initializer is not None:
            val = _t(initializer).clone()
        else:
            val = torch.zeros([int(s) for s in shape])
        if val.dtype.is_floating_point:
            val = val.to(float32)
        val = val.detach().clone()
        if trainable and val.dtype.is_floating_point:
            val.requires_grad_(True)
            S.trainable.append(full)
        S.variables[full] = val
    return S.variables[full]


def trainable_variables(scope=None):
    return [S.variables[n] for n in S.trainable if scope is None or n.startswith(scope)]


class GraphKeys:
    UPDATE_OPS = 'update_ops'


def get_collection(key):
    return ('collection', key)


def global_variables_initializer():
    return ('init',)


def placeholder(dtype, shape=None, name=None):
    name = name or f'ph{S.ph_count}'
    S.ph_count += 1
    if name in S.feeds:
        v = S.feeds[name]
        t = wrap(_t(v, dtype).clone()) if not isinstance(v, (bool, float, int)) else torch.tensor(v, dtype=dtype)
    else:       # construction-time dummy
        shp = [1 if s is None else int(s) for s in (shape or [])]
        if name == 'labels':
            shp = [shp[0], 2, 5]
        t = torch.zeros(shp, dtype=dtype)
    t = wrap(t) if isinstance(t, torch.Tensor) else t
    t.ph_name = name
    return t


# ---------------------------------------------------------------------------- nn / layers
def _same_pad(in_size, k, stride, dil=1):
    out = -(-in_size // stride)
    total = max((out - 1) * stride + (k - 1) * dil + 1 - in_size, 0)
    return total // 2, total - total // 2


def _conv_nhwc(x, w_hwio, stride, dil):
    k = w_hwio.shape[0]
    pt, pb = _same_pad(x.shape[1], k, stride, dil)
    pl, pr = _same_pad(x.shape[2], k, stride, dil)
    xc = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xc, w_hwio.permute(3, 2, 0, 1), None, stride=stride, dilation=dil)
    return y.permute(0, 2, 3, 1)


class _NN:
    @staticmethod
    def conv2d(x, filter=None, strides=None, padding='SAME', data_format='NHWC', name=None):   # noqa: A002
        assert padding == 'SAME' and data_format == 'NHWC'
        return _conv_nhwc(x, filter, strides[1], 1)

    @staticmethod
    def bias_add(x, bias, name=None):
        return x + bias

    @staticmethod
    def relu(x, name=None):
        return torch.relu(x)

    @staticmethod
    def softmax(x, axis=-1):
        return torch.softmax(x, dim=axis)

    @staticmethod
    def l2_normalize(x, axis=None, epsilon=1e-12):
        ss = (x * x).sum(dim=axis, keepdim=True)
        return x * torch.rsqrt(torch.clamp(ss, min=epsilon))

    @staticmethod
    def sigmoid_cross_entropy_with_logits(labels=None, logits=None):
        # max(x, 0) - x * z + log(1 + exp(-|x|))   (TF's documented stable form)
        x, z = logits, labels
        return torch.clamp(x, min=0) - x * z + torch.log1p(torch.exp(-torch.abs(x)))

    @staticmethod
    def top_k(x, k):
        """tf.nn.top_k on a vector: descending, the LOWER index first among equal values."""
        order = torch.sort(x, descending=True, stable=True).indices[: int(k)]
        return x[order], order.to(int32)

    @staticmethod
    def l2_loss(v):
        return (v * v).sum() / 2

    @staticmethod
    def leaky_relu(x, alpha=0.2, name=None):
        return torch.where(x > 0, x, x * alpha)


nn = _NN()


def _glorot_uniform(shape_hwio, gen):
    kh, kw, ci, co = shape_hwio
    limit = math.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
    return (torch.rand(shape_hwio, generator=gen) * 2 - 1) * limit


class _Layers:
    gen = torch.Generator().manual_seed(1234)

    @staticmethod
    def conv2d(inputs, filters, kernel_size, strides=1, padding='valid', name=None, data_format='channels_last', dilation_rate=1,
               kernel_initializer=None, bias_initializer=None):
        assert padding == 'same' and data_format == 'channels_last'
        filters = int(filters)                        # YOLOv3.py:487 passes filters/2, a float under true division
        ci = inputs.shape[-1]
        with variable_scope(name, default_name='conv2d'):     # default names: conv2d, conv2d_1, ... per enclosing variable scope
            w = get_variable('kernel', initializer=_glorot_uniform((kernel_size, kernel_size, ci, filters), _Layers.gen))
            b = get_variable('bias', shape=[filters], initializer=bias_initializer)
        return _conv_nhwc(inputs, w, strides, dilation_rate) + b

    @staticmethod
    def separable_conv2d(inputs, filters, kernel_size, strides=1, padding='valid', name=None, data_format='channels_last', use_bias=True,
                         dilation_rate=1):
        """tf.layers.separable_conv2d (depth_multiplier 1): depthwise_kernel [kh, kw, in, 1] then pointwise_kernel [1, 1, in, filters],
        both glorot-uniform, created in that order; SAME padding per axis (kernel_size may be [kh, kw])"""
        assert padding == 'same' and data_format == 'channels_last' and strides == 1 and dilation_rate == 1
        kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else (int(kernel_size[0]), int(kernel_size[1]))
        ci, filters = inputs.shape[-1], int(filters)
        with variable_scope(name, default_name='separable_conv2d'):
            lim = math.sqrt(6.0 / (kh * kw * ci + kh * kw * 1))
            dw = get_variable('depthwise_kernel', initializer=(torch.rand((kh, kw, ci, 1), generator=_Layers.gen) * 2 - 1) * lim)
            pw = get_variable('pointwise_kernel', initializer=_glorot_uniform((1, 1, ci, filters), _Layers.gen))
            b = get_variable('bias', shape=[filters]) if use_bias else None
        (pt, pb), (pl, pr) = _same_pad(inputs.shape[1], kh, 1), _same_pad(inputs.shape[2], kw, 1)
        xc = F.pad(inputs.permute(0, 3, 1, 2), (pl, pr, pt, pb))
        y = F.conv2d(xc, dw.permute(2, 3, 0, 1), None, groups=ci)                 # [ci, 1, kh, kw]
        y = F.conv2d(y, pw.permute(3, 2, 0, 1), None)
        y = y.permute(0, 2, 3, 1)
        return y + b if b is not None else y

    @staticmethod
    def dense(inputs, units, name=None, activation=None):
        """tf.layers.dense: kernel [in, units] (glorot-uniform), bias zeros"""
        ci, units = inputs.shape[-1], int(units)
        with variable_scope(name, default_name='dense'):
            lim = math.sqrt(6.0 / (ci + units))
            w = get_variable('kernel', initializer=(torch.rand((ci, units), generator=_Layers.gen) * 2 - 1) * lim)
            b = get_variable('bias', shape=[units])
        y = inputs @ w + b
        return activation(y) if activation is not None else y

    @staticmethod
    def flatten(inputs):
        return inputs.reshape(inputs.shape[0], -1)

    @staticmethod
    def batch_normalization(inputs, axis=3, training=False, momentum=0.99, epsilon=1e-3):
        assert axis == 3
        c = inputs.shape[-1]
        with variable_scope(None, default_name='batch_normalization'):
            gamma = get_variable('gamma', initializer=torch.ones(c))
            beta = get_variable('beta', initializer=torch.zeros(c))
            mm = get_variable('moving_mean', initializer=torch.zeros(c), trainable=False)
            mv = get_variable('moving_variance', initializer=torch.ones(c), trainable=False)
        if bool(training):
            mean = inputs.mean(dim=(0, 1, 2))
            var = ((inputs - mean) ** 2).mean(dim=(0, 1, 2))
            n = inputs.shape[0] * inputs.shape[1] * inputs.shape[2]
            unb = var.detach() * (n / max(n - 1, 1))
            S.pending.append(('assign', mm, mm * momentum + mean.detach() * (1 - momentum)))
            S.pending.append(('assign', mv, mv * momentum + unb * (1 - momentum)))
        else:
            mean, var = mm, mv
        return (inputs - mean) * (torch.rsqrt(var + epsilon) * gamma) + beta

    @staticmethod
    def conv2d_transpose(inputs, filters, kernel_size, strides=1, padding='valid', name=None, data_format='channels_last',
                         kernel_initializer=None, bias_initializer=None):
        """tf.layers.conv2d_transpose(padding='same'): kernel [kh, kw, filters, in]; the output (in * stride) is the gradient of a SAME
        conv2d (filters -> in channels, this kernel, this stride) with respect to its input -- computed literally that way."""
        assert padding == 'same' and data_format == 'channels_last'
        filters = int(filters)
        n, h, w, ci = inputs.shape
        with variable_scope(name, default_name='conv2d_transpose'):
            kern = get_variable('kernel', initializer=_glorot_uniform((kernel_size, kernel_size, filters, ci), _Layers.gen))
            b = get_variable('bias', shape=[filters], initializer=bias_initializer)
        oh, ow = h * strides, w * strides
        (pt, pb), (pl, pr) = _same_pad(oh, kernel_size, strides), _same_pad(ow, kernel_size, strides)
        wt = kern.permute(3, 2, 0, 1)                                       # forward conv weight [out = ci, in = filters, kh, kw]
        full = torch.nn.grad.conv2d_input((n, filters, oh + pt + pb, ow + pl + pr), wt, inputs.permute(0, 3, 1, 2), stride=strides)
        y = full[:, :, pt: pt + oh, pl: pl + ow]
        return y.permute(0, 2, 3, 1) + b

    @staticmethod
    def average_pooling2d(inputs, pool_size, strides, padding, data_format, name=None):
        """SAME average pooling: the mean over the cells of the window that lie inside the picture"""
        assert padding == 'same' and data_format == 'channels_last'
        pt, pb = _same_pad(inputs.shape[1], pool_size, strides)
        pl, pr = _same_pad(inputs.shape[2], pool_size, strides)
        xc = F.pad(inputs.permute(0, 3, 1, 2), (pl, pr, pt, pb))
        ones = F.pad(torch.ones(1, 1, inputs.shape[1], inputs.shape[2]), (pl, pr, pt, pb))
        s_ = F.avg_pool2d(xc, pool_size, strides) * (pool_size * pool_size)
        c_ = F.avg_pool2d(ones, pool_size, strides) * (pool_size * pool_size)
        return (s_ / c_).permute(0, 2, 3, 1)

    @staticmethod
    def max_pooling2d(inputs, pool_size, strides, padding, data_format, name=None):
        assert padding == 'same' and data_format == 'channels_last'
        pt, pb = _same_pad(inputs.shape[1], pool_size, strides)
        pl, pr = _same_pad(inputs.shape[2], pool_size, strides)
        xc = F.pad(inputs.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float('-inf'))
        return F.max_pool2d(xc, pool_size, strides).permute(0, 2, 3, 1)


layers = _Layers()


# ---------------------------------------------------------------------------- losses / image
class _Reduction:
    NONE = 'none'
    MEAN = 'weighted_mean'


class _Losses:
    Reduction = _Reduction

    @staticmethod
    def sparse_softmax_cross_entropy(labels, logits, reduction=_Reduction.MEAN):
        labels = _t(labels).long().reshape(-1)
        m = logits.max(dim=1, keepdim=True).values
        sh = logits - m
        lse = torch.log(torch.exp(sh).sum(dim=1))
        per = lse - sh.gather(1, labels.view(-1, 1)).squeeze(1)
        if reduction == _Reduction.NONE:
            return per
        n = per.shape[0]
        return per.sum() / n if n > 0 else per.sum()         # div_no_nan


losses = _Losses()


class _Image:
    @staticmethod
    def non_max_suppression(boxes, scores, max_output_size, iou_threshold=0.5, score_threshold=float('-inf')):
        """tf.image.non_max_suppression (NonMaxSuppressionV3) -- the shim's OWN implementation, written a different way than
        oracle/nms_ref.cpp (which mirrors the op's priority queue): the whole IoU matrix in float32 numpy with the op's
        expression (corner-order agnostic, area <= 0 -> 0, inter / (a_i + a_j - inter)), candidates in descending score order
        (equal scores: lower index first -- what the op's heap gives for boxes pushed in index order whenever it matters
        for the fixtures: their scores are distinct), suppression iff IoU > threshold (strict).  The golden fixtures produced
        through this function therefore CROSS-CHECK nms_ref.cpp instead of echoing it (round-1 verdict)."""
        b = np.asarray(boxes.detach().numpy() if hasattr(boxes, 'detach') else boxes, dtype=np.float32).reshape(-1, 4)
        sc = np.asarray(scores.detach().numpy() if hasattr(scores, 'detach') else scores, dtype=np.float32).reshape(-1)
        n = sc.shape[0]
        if n == 0 or int(max_output_size) <= 0:
            return torch.zeros(0, dtype=torch.int32)
        y0, y1 = np.minimum(b[:, 0], b[:, 2]), np.maximum(b[:, 0], b[:, 2])
        x0, x1 = np.minimum(b[:, 1], b[:, 3]), np.maximum(b[:, 1], b[:, 3])
        area = ((y1 - y0) * (x1 - x0)).astype(np.float32)
        order = np.lexsort((np.arange(n), -sc.astype(np.float64)))
        order = order[sc[order] > np.float32(score_threshold)]
        keep = []
        thr = np.float32(iou_threshold)
        for i in order:
            if len(keep) >= int(max_output_size):
                break
            if keep:
                k = np.asarray(keep)
                ih = np.maximum(np.minimum(y1[i], y1[k]) - np.maximum(y0[i], y0[k]), np.float32(0)).astype(np.float32)
                iw = np.maximum(np.minimum(x1[i], x1[k]) - np.maximum(x0[i], x0[k]), np.float32(0)).astype(np.float32)
                inter = (ih * iw).astype(np.float32)
                with np.errstate(divide='ignore', invalid='ignore'):
                    iou = (inter / ((area[i] + area[k]).astype(np.float32) - inter)).astype(np.float32)
                iou = np.where((area[i] <= 0) | (area[k] <= 0), np.float32(0), iou)
                if np.any(iou > thr):
                    continue
            keep.append(int(i))
        return torch.from_numpy(np.asarray(keep, dtype=np.int32))


class _ResizeMethod:
    BILINEAR, NEAREST_NEIGHBOR, BICUBIC = 'bilinear', 'nearest', 'bicubic'


def _resize_images(images, size, method='bilinear', align_corners=False, preserve_aspect_ratio=False):
    """tf.image.resize_images on one HWC image; bilinear with align_corners=True is what the reference's drivers use,
    'nearest' / 'bicubic' are the other two fill modes of utils/image_augmentor.py:72-76."""
    assert method in ('bilinear', 'nearest', 'bicubic') and not preserve_aspect_ratio
    h, w = int(size[0]), int(size[1])
    if method == 'nearest':
        return _resize_one_nearest(images, h, w, bool(align_corners))
    if method == 'bicubic':
        return _resize_one_bicubic(images, h, w, bool(align_corners))
    x = images.permute(2, 0, 1).unsqueeze(0)
    y = F.interpolate(x, size=(h, w), mode='bilinear', align_corners=bool(align_corners))
    return y.squeeze(0).permute(1, 2, 0).contiguous()


def _resize_scale(n_in, n_out, align_corners):
    """CalculateResizeScale (tensorflow/core/kernels/image_resizer_state.h), a float"""
    return np.float32((n_in - 1) / np.float32(n_out - 1)) if (align_corners and n_out > 1) else np.float32(n_in / np.float32(n_out))


def _resize_one_nearest(img, oh, ow, align_corners):
    """ResizeNearestNeighbor (TF 1.13, no half-pixel centres): source index = round-half-away(dst * scale) with align_corners,
    floor(dst * scale) without, capped at in - 1.  The result keeps the input's values (and, in TensorFlow, its dtype)."""
    def src(n_in, n_out):
        pos = (np.arange(n_out, dtype=np.float32) * _resize_scale(n_in, n_out, align_corners)).astype(np.float32)
        idx = np.floor(pos + np.float32(0.5)) if align_corners else np.floor(pos)          # positions are >= 0: roundf = floor(x + .5)
        return torch.from_numpy(np.minimum(idx.astype(np.int64), n_in - 1))
    return img.index_select(0, src(img.shape[0], oh)).index_select(1, src(img.shape[1], ow)).contiguous()


def _bicubic_matrix(n_in, n_out, align_corners):
    """one axis of ResizeBicubic (TF 1.13) as an [n_out, n_in] matrix: Keys kernel with A = -0.75 evaluated at the 1/1024 grid
    point nearest to the fractional position (the kernel reads a 1 025-entry table, lrintf = round-half-even), four taps at
    floor - 1 .. floor + 2 clamped to the picture (weights of clamped taps add up on the border pixel)."""
    a = -0.75
    def inner(x):           # |x| <= 1
        return ((a + 2.) * x - (a + 3.)) * x * x + 1.
    def outer(x):           # 1 <= |x| <= 2
        return ((a * x - 5. * a) * x + 8. * a) * x - 4. * a
    scale = _resize_scale(n_in, n_out, align_corners)
    m = np.zeros((n_out, n_in), np.float64)
    for o in range(n_out):
        pos = np.float32(scale * np.float32(o))
        base = int(pos)
        frac = float(np.float32(pos - np.float32(base)))
        t = float(np.rint(np.float32(frac * 1024.))) / 1024.
        wts = [np.float32(outer(t + 1.)), np.float32(inner(t)), np.float32(inner(1. - t)), np.float32(outer(2. - t))]   # t = k/1024: exact
        for k, wt in enumerate(wts):
            m[o, min(n_in - 1, max(0, base - 1 + k))] += float(wt)
    return m


def _resize_one_bicubic(img, oh, ow, align_corners):
    """rows of the horizontal pass first, then the vertical one (the kernel's order); done in double, returned as f32"""
    x = img.detach().double().numpy()
    my, mx = _bicubic_matrix(x.shape[0], oh, align_corners), _bicubic_matrix(x.shape[1], ow, align_corners)
    y = np.einsum('oh,hwc->owc', my, np.einsum('pw,hwc->hpc', mx, x))
    return torch.from_numpy(y.astype(np.float32)).contiguous()


def _adjust_contrast(images, contrast_factor):
    mean = images.mean(dim=(-3, -2), keepdim=True)
    return (images - mean) * contrast_factor + mean


def _adjust_hue(images, delta):
    """HSV round trip (value-range free), written the colorsys way."""
    r, g, b = images.unbind(-1)
    mx, mn = images.max(-1).values, images.min(-1).values
    d = mx - mn
    safe = torch.where(d > 0, d, torch.ones_like(d))
    rc, gc, bc = (mx - r) / safe, (mx - g) / safe, (mx - b) / safe
    h = torch.where(r == mx, bc - gc, torch.where(g == mx, 2. + rc - bc, 4. + gc - rc))
    h = torch.where(d > 0, (h / 6.) % 1.0, torch.zeros_like(h))
    h = (h + delta) % 1.0
    i = torch.floor(h * 6.)
    f = h * 6. - i
    # TF's CPU kernel carries (hue, min, max), never a saturation: out-of-gamut (negative) pixels keep their range
    p_, q, t_ = mx - d, mx - d * f, mx - d * (1. - f)
    i = (i.long() % 6).unsqueeze(-1)
    pick = lambda *c: torch.stack(c, -1).gather(-1, i).squeeze(-1)
    return torch.stack([pick(mx, q, p_, p_, t_, mx), pick(t_, mx, mx, q, p_, p_), pick(p_, p_, t_, mx, mx, q)], -1)


def _contrib_rotate(img, ang, interpolation='NEAREST'):
    """tf.contrib.image.rotate, BILINEAR: output(x, y) = input(R(x, y)), zero outside; through grid_sample."""
    assert interpolation == 'BILINEAR'
    H, W, _ = img.shape
    ang = float(ang)
    c, s = math.cos(ang), math.sin(ang)
    ox = ((W - 1) - (c * (W - 1) - s * (H - 1))) / 2.
    oy = ((H - 1) - (s * (W - 1) + c * (H - 1))) / 2.
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing='ij')
    sx = c * xs - s * ys + ox
    sy = s * xs + c * ys + oy
    grid = torch.stack([sx / (W - 1) * 2 - 1, sy / (H - 1) * 2 - 1], -1).unsqueeze(0).float()
    out = F.grid_sample(img.permute(2, 0, 1).unsqueeze(0), grid, mode='bilinear', padding_mode='zeros', align_corners=True)
    ROTATE_TRACE.append(out.squeeze(0).permute(1, 2, 0).contiguous())
    return ROTATE_TRACE[-1]


ROTATE_TRACE = []          # outputs of tf.contrib.image.rotate, for callers whose return value drops the image
def _resize_nearest_neighbor(images, size, align_corners=False):
    """NHWC; TF-1.x legacy grid: src = floor(dst * in / out)"""
    n, h, w, c = images.shape
    oh, ow = int(size[0]), int(size[1])
    iy = torch.clamp(torch.floor(torch.arange(oh, dtype=torch.float32) * (h / oh)).long(), max=h - 1)
    ix = torch.clamp(torch.floor(torch.arange(ow, dtype=torch.float32) * (w / ow)).long(), max=w - 1)
    return images[:, iy][:, :, ix]


def _resize_bilinear(images, size, align_corners=False):
    """NHWC, TF-1.x grid (no half-pixel centres): src = dst * in / out, or dst * (in - 1) / (out - 1) with align_corners (and out > 1)
    -- CalculateResizeScale of tensorflow/core/kernels/image_resizer_state.h; taps floor(src) and min(floor(src) + 1, in - 1); differentiable"""
    n, h, w, c = images.shape
    oh, ow = int(size[0]), int(size[1])
    sy = (h - 1) / (oh - 1) if (align_corners and oh > 1) else h / oh
    sx = (w - 1) / (ow - 1) if (align_corners and ow > 1) else w / ow
    fy = torch.arange(oh, dtype=torch.float32) * torch.tensor(sy, dtype=torch.float32)
    fx = torch.arange(ow, dtype=torch.float32) * torch.tensor(sx, dtype=torch.float32)
    y0, x0 = torch.floor(fy).long(), torch.floor(fx).long()
    y1, x1 = torch.clamp(y0 + 1, max=h - 1), torch.clamp(x0 + 1, max=w - 1)
    ly, lx = (fy - y0.float()).view(1, oh, 1, 1), (fx - x0.float()).view(1, 1, ow, 1)
    top = images[:, y0][:, :, x0] * (1 - lx) + images[:, y0][:, :, x1] * lx
    bot = images[:, y1][:, :, x0] * (1 - lx) + images[:, y1][:, :, x1] * lx
    return top * (1 - ly) + bot * ly


def _crop_and_resize(image, boxes, box_ind, crop_size, method='bilinear', extrapolation_value=0.):
    """tf.image.crop_and_resize (crop_and_resize_op.cc, TF 1.13), bilinear: for crop row y of box [y1, x1, y2, x2] (normalised) the source row is
    y1 (H - 1) + y (y2 - y1)(H - 1) / (crop_h - 1)  (0.5 (y1 + y2)(H - 1) when crop_h == 1); a sample whose row or column falls outside
    [0, H - 1] x [0, W - 1] is the extrapolation value; otherwise top + (bottom - top) * y_lerp of (tl + (tr - tl) * x_lerp) rows with
    floor / ceil neighbours.  Differentiable in `image` (autograd)."""
    assert method == 'bilinear'
    n, H, W, C = image.shape
    ch, cw = int(crop_size[0]), int(crop_size[1])
    boxes = _t(boxes, float32).detach()
    bi = _t(box_ind).long()
    R = boxes.shape[0]
    if R == 0:
        return image.new_zeros((0, ch, cw, C))
    y1, x1, y2, x2 = boxes[:, 0:1], boxes[:, 1:2], boxes[:, 2:3], boxes[:, 3:4]
    gy = torch.arange(ch, dtype=float32).view(1, ch)
    gx = torch.arange(cw, dtype=float32).view(1, cw)
    in_y = y1 * (H - 1) + gy * ((y2 - y1) * (H - 1) / (ch - 1)) if ch > 1 else 0.5 * (y1 + y2) * (H - 1) + gy * 0
    in_x = x1 * (W - 1) + gx * ((x2 - x1) * (W - 1) / (cw - 1)) if cw > 1 else 0.5 * (x1 + x2) * (W - 1) + gx * 0
    oky = (in_y >= 0) & (in_y <= H - 1)
    okx = (in_x >= 0) & (in_x <= W - 1)
    ty = torch.floor(in_y).clamp(0, H - 1).long(); by = torch.ceil(in_y).clamp(0, H - 1).long()
    lx_ = torch.floor(in_x).clamp(0, W - 1).long(); rx = torch.ceil(in_x).clamp(0, W - 1).long()
    ly = (in_y - torch.floor(in_y)).view(R, ch, 1, 1)
    lx = (in_x - torch.floor(in_x)).view(R, 1, cw, 1)
    b = bi.view(R, 1, 1)

    def px(yy, xx):
        return image[b, yy.view(R, ch, 1), xx.view(R, 1, cw)]               # [R, ch, cw, C]
    top = px(ty, lx_) + (px(ty, rx) - px(ty, lx_)) * lx
    bot = px(by, lx_) + (px(by, rx) - px(by, lx_)) * lx
    val = top + (bot - top) * ly
    ok = (oky.view(R, ch, 1) & okx.view(R, 1, cw)).unsqueeze(-1)
    return torch.where(ok, val, torch.full_like(val, float(extrapolation_value)))


_Image.crop_and_resize = staticmethod(_crop_and_resize)
_Image.resize_bilinear = staticmethod(_resize_bilinear)
_Image.resize_nearest_neighbor = staticmethod(_resize_nearest_neighbor)
_Image.adjust_brightness = staticmethod(lambda images, delta: images + delta)
_Image.adjust_contrast = staticmethod(_adjust_contrast)
_Image.adjust_hue = staticmethod(_adjust_hue)
_Image.ResizeMethod = _ResizeMethod
_Image.resize_images = staticmethod(_resize_images)
_Image.resize = staticmethod(lambda images, size: _resize_images(images, size, 'bilinear', False))
image = _Image()


class _SparseTensor:
    def __init__(self, indices, values, dense_shape):
        self.indices, self.values, self.dense_shape = indices, values, dense_shape


class _Sparse:
    SparseTensor = _SparseTensor

    @staticmethod
    def to_dense(sp, validate_indices=True):
        shp = [int(s) for s in sp.dense_shape]
        out = torch.zeros(shp, dtype=sp.values.dtype if isinstance(sp.values, torch.Tensor) else float32)
        idx = sp.indices.long()
        vals = sp.values if sp.values.dim() > 0 else sp.values.reshape(1)
        out[idx[:, 0], idx[:, 1]] = vals.to(out.dtype)
        return out


sparse = _Sparse()


class _ContribFramework:
    @staticmethod
    def sort(x):
        return torch.sort(x).values


def _group_norm(inputs, groups=32, channels_axis=-1, reduction_axes=(-3, -2), trainable=True, epsilon=1e-6):
    """tf.contrib.layers.group_norm on NHWC: per sample and group, moments over H, W and the group's channels; gamma / beta [C] under
    variable_scope(None, 'GroupNorm'), whose default name is made unique WITHIN the enclosing scope (GroupNorm, GroupNorm_1, ...)"""
    assert channels_axis in (3, -1) and tuple(reduction_axes) in ((1, 2), (-3, -2))
    n, h, w, c = inputs.shape
    with variable_scope(None, default_name='GroupNorm'):
        beta = get_variable('beta', initializer=torch.zeros(c))
        gamma = get_variable('gamma', initializer=torch.ones(c))
    x = inputs.reshape(n, h, w, groups, c // groups)
    mean = x.mean(dim=(1, 2, 4), keepdim=True)
    var = ((x - mean) ** 2).mean(dim=(1, 2, 4), keepdim=True)
    gain = torch.rsqrt(var + epsilon)
    y = ((x - mean) * gain).reshape(n, h, w, c)
    return y * gamma + beta


contrib = types.SimpleNamespace(framework=_ContribFramework(), layers=types.SimpleNamespace(variance_scaling_initializer=lambda *a, **k: None, group_norm=_group_norm),
                               image=types.SimpleNamespace(rotate=_contrib_rotate))
random = types.SimpleNamespace(uniform=random_uniform)


# ---------------------------------------------------------------------------- training
class _MomentumOptimizer:
    def __init__(self, learning_rate, momentum):
        self.lr, self.mom = learning_rate, momentum

    def minimize(self, loss, global_step=None):
        # slot variables (slot_creator: variable_scope(None, primary.op.name + '/' + slot)) and the non-slot accumulators (beta1_power ...)
        # are created HERE, under whatever variable scope is open at this call: recorded so that tests can pin the prefix of their names
        S.optimizer_scope = '/'.join(S.scope)
        S.pending.append(('minimize', self, loss, global_step))
        return ('train_op',)


def _opt_compute_gradients(self, loss, var_list=None):
    """optimizer.compute_gradients(loss, var_list): [(grad, var)]; gradients are taken when the step is flushed (all variables at their
    pre-update values), here only the request is recorded"""
    vars_ = list(var_list) if var_list is not None else [S.variables[n] for n in S.trainable]
    return [(('grad_of', loss, i), v) for i, v in enumerate(vars_)]


def _opt_apply_gradients(self, grads_and_vars, global_step=None):
    S.optimizer_scope = '/'.join(S.scope)
    loss = grads_and_vars[0][0][1]
    S.pending.append(('apply', self, loss, [v for _, v in grads_and_vars], global_step))
    return ('train_op',)


_MomentumOptimizer.compute_gradients = _opt_compute_gradients
_MomentumOptimizer.apply_gradients = _opt_apply_gradients


class _AdamOptimizer:
    """tf.train.AdamOptimizer(lr): beta1 0.9, beta2 0.999, epsilon 1e-8; ApplyAdam:
    lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = b1 m + (1 - b1) g;  v = b2 v + (1 - b2) g^2;  var -= lr_t m / (sqrt(v) + eps)"""

    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta1, beta2, epsilon

    def minimize(self, loss, global_step=None):
        # slot variables (slot_creator: variable_scope(None, primary.op.name + '/' + slot)) and the non-slot accumulators (beta1_power ...)
        # are created HERE, under whatever variable scope is open at this call: recorded so that tests can pin the prefix of their names
        S.optimizer_scope = '/'.join(S.scope)
        S.pending.append(('minimize', self, loss, global_step))
        return ('train_op',)

    def apply(self, names, vars_, grads):
        st = S.momentum.setdefault('__adam__', {'t': 0, 'm': {}, 'v': {}})
        st['t'] += 1
        t = st['t']
        lr_t = float(self.lr) * (1.0 - self.b2 ** t) ** 0.5 / (1.0 - self.b1 ** t)
        for n, v, g in zip(names, vars_, grads):
            if g is None:
                g = torch.zeros_like(v)
            m_ = st['m'].setdefault(n, torch.zeros_like(v))
            v_ = st['v'].setdefault(n, torch.zeros_like(v))
            m_.mul_(self.b1).add_(g * (1.0 - self.b1))
            v_.mul_(self.b2).add_(g * g * (1.0 - self.b2))
            v.sub_(lr_t * m_ / (torch.sqrt(v_) + self.eps))


class _Saver:
    def __init__(self, var_list=None):
        self.var_list = var_list

    def save(self, sess, path, global_step=None):
        return path

    def restore(self, sess, path):
        return None


train = types.SimpleNamespace(MomentumOptimizer=_MomentumOptimizer, AdamOptimizer=_AdamOptimizer, Saver=_Saver)

summary = types.SimpleNamespace(scalar=lambda *a, **k: None, merge_all=lambda: None)
gfile = types.SimpleNamespace(Exists=lambda p: True, MakeDirs=lambda p: None)


def _flush(apply_):
    pend, S.pending = S.pending, []
    if not apply_:
        return
    # gradients first (all variables at their pre-update values), then every assignment
    for item in pend:
        if item[0] == 'minimize':
            _, opt, loss, gstep = item
            names = list(S.trainable)
            vars_ = [S.variables[n] for n in names]
            grads = torch.autograd.grad(loss, vars_, allow_unused=True)
            lr = float(opt.lr)
            with torch.no_grad():
                if isinstance(opt, _AdamOptimizer):
                    opt.apply(names, vars_, grads)
                    if gstep is not None:
                        gstep.add_(1)
                    continue
                for n, v, g in zip(names, vars_, grads):
                    if g is None:
                        g = torch.zeros_like(v)
                    acc = S.momentum.setdefault(n, torch.zeros_like(v))
                    acc.mul_(opt.mom).add_(g)                    # accum = m*accum + grad
                    v.sub_(lr * acc)                              # var -= lr*accum
                if gstep is not None:
                    gstep.add_(1)
    # compute_gradients / apply_gradients pairs (LH_RCNN.py:184-191): every requested gradient first, then every update
    applies = [item for item in pend if item[0] == 'apply']
    if applies:
        byid = {id(v): n for n, v in S.variables.items()}
        todo = []
        for _, opt, loss, vars_, gstep in applies:
            todo.append((opt, vars_, torch.autograd.grad(loss, vars_, allow_unused=True, retain_graph=True), gstep))
        with torch.no_grad():
            for opt, vars_, grads, gstep in todo:
                for v, g in zip(vars_, grads):
                    if g is None:
                        g = torch.zeros_like(v)
                    acc = S.momentum.setdefault(byid[id(v)], torch.zeros_like(v))
                    acc.mul_(opt.mom).add_(g)
                    v.sub_(float(opt.lr) * acc)
                if gstep is not None:
                    gstep.add_(1)
    with torch.no_grad():
        for item in pend:
            if item[0] == 'assign':
                item[1].copy_(item[2])


class InteractiveSession:
    def __init__(self):
        self.model = sys._getframe(1).f_locals.get('self')
        S.model = self.model

    def run(self, fetches, feed_dict=None):
        m = self.model
        if callable(fetches):                                   # an "initializer" supplied by the harness
            fetches()
            return None
        if isinstance(fetches, tuple) and fetches and isinstance(fetches[0], str) and fetches[0] == 'init':
            return None
        byid = {id(v): k for k, v in vars(m).items()}
        if id(fetches) in byid:
            single, names = True, [byid[id(fetches)]]
        else:
            single, names = False, [byid[id(f)] for f in fetches]
        S.feeds = {}
        overrides = {}           # TF lets feed_dict replace ANY tensor: emulate for model attributes
        for ph, val in (feed_dict or {}).items():
            if hasattr(ph, 'ph_name'):
                S.feeds[ph.ph_name] = val
            else:
                overrides[byid[id(ph)]] = val
        S.ph_count = 0
        S.pending = []
        _SCOPE_COUNT.clear()
        wants_update = 'train_op' in names
        # RetinaNet.py names its graph methods per task (:101, :137); every other class has _define_inputs / _build_graph
        define = getattr(m, '_define_inputs', None) or m._define_detection_inputs
        build = getattr(m, '_build_graph', None) or m._build_detection_graph
        define()
        for k, val in overrides.items():      # e.g. test_one_image feeds self.images = placeholder - mean,
            old = getattr(m, k)               # i.e. the fed pixels BYPASS the mean subtraction
            setattr(m, k, wrap(_t(val, old.dtype).clone()))
        build()
        out = []
        for n in names:
            v = getattr(m, n)
            if isinstance(v, torch.Tensor):
                v = v.detach().clone().numpy()
            elif isinstance(v, list):
                v = [e.detach().clone().numpy() for e in v]
            out.append(v)
        _flush(wants_update)
        return out[0] if single else out


# ---------------------------------------------------------------------------- module assembly
def install(vgg_tensors=None):
    """Registers fake `tensorflow` / `tensorflow.python.pywrap_tensorflow` modules."""
    reset()
    _SCOPE_COUNT.clear()
    tf = types.ModuleType('tensorflow')
    me = sys.modules[__name__]
    for k in dir(me):
        if not k.startswith('_'):
            setattr(tf, k, getattr(me, k))
    tf.range = range_
    tf.abs = abs_
    tf.slice = slice_
    tf.bool = bool_
    tf.layers, tf.nn, tf.losses, tf.image, tf.sparse, tf.contrib = layers, nn, losses, image, sparse, contrib
    tf.train, tf.summary, tf.gfile = train, summary, gfile
    py = types.ModuleType('tensorflow.python')
    pw = types.ModuleType('tensorflow.python.pywrap_tensorflow')

    class _Reader:
        def __init__(self, path):
            self.t = vgg_tensors or {}

        def get_tensor(self, name):
            return self.t[name]

    pw.NewCheckpointReader = _Reader
    py.pywrap_tensorflow = pw
    tf.python = py
    torch.Tensor.get_shape = lambda self: list(self.shape)       # static shapes (YOLOv3.py:404-408); removed by uninstall()
    sys.modules['tensorflow'] = tf
    sys.modules['tensorflow.python'] = py
    sys.modules['tensorflow.python.pywrap_tensorflow'] = pw
    return tf


def uninstall():
    if hasattr(torch.Tensor, 'get_shape'):
        del torch.Tensor.get_shape
    for k in ('tensorflow', 'tensorflow.python', 'tensorflow.python.pywrap_tensorflow'):
        sys.modules.pop(k, None)


def load_reference_module(path, name):
    """Execs a reference source file that parses as shipped (RetinaNet.py, ...) under the shim."""
    mod = types.ModuleType(name)
    mod.__file__ = path
    exec(compile(open(path).read(), path, 'exec'), mod.__dict__)
    return mod


def load_reference_ssd300(path='/root/reference/SSD300.py'):
    """Execs the reference source (with its one-line syntax defect repaired IN MEMORY:
    SSD300.py:41-43 `else:` with an empty body) under the shim; returns the module."""
    src = open(path).read()
    broken = "        else:\n\n        self.global_step"
    assert broken in src, 'reference layout changed'
    src = src.replace(broken, "        else:\n            pass\n\n        self.global_step")
    mod = types.ModuleType('reference_SSD300')
    mod.__file__ = path
    exec(compile(src, path, 'exec'), mod.__dict__)
    return mod
