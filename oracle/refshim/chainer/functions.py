"""`chainer.functions` stand-in (only what the reference calls), numpy forward values + analytic
vector-Jacobian products written with these same functions (so gradients are differentiable again)."""
import numpy as np

from .variable import Variable, _raw, make, make_multi


def _shape(x):
    return np.shape(_raw(x))


def sum_to(g, shape):
    """undo numpy broadcasting: reduce `g` to `shape`"""
    shape = tuple(shape)
    if g.shape == shape:
        return g
    lead = g.ndim - len(shape)
    axes = tuple(range(lead)) + tuple(lead + i for i, s in enumerate(shape) if s == 1 and g.shape[lead + i] != 1)
    out = sum(g, axis=axes, keepdims=True) if axes else g
    return reshape(out, shape)


def broadcast_to(x, shape):
    shape = tuple(shape)
    if _shape(x) == shape:
        return x if isinstance(x, Variable) else Variable(x)
    xs = _shape(x)
    return make(np.broadcast_to(_raw(x), shape), (x,), lambda g: (sum_to(g, xs),))


# ---- arithmetic (Variable operators)
def add(a, b):
    sa, sb = _shape(a), _shape(b)
    return make(_raw(a) + _raw(b), (a, b), lambda g: (sum_to(g, sa), sum_to(g, sb)))


def sub(a, b):
    sa, sb = _shape(a), _shape(b)
    return make(_raw(a) - _raw(b), (a, b), lambda g: (sum_to(g, sa), sum_to(neg(g), sb)))


def mul(a, b):
    sa, sb = _shape(a), _shape(b)
    return make(_raw(a) * _raw(b), (a, b), lambda g: (sum_to(g * b, sa), sum_to(g * a, sb)))


def div(a, b):
    sa, sb = _shape(a), _shape(b)
    return make(_raw(a) / _raw(b), (a, b), lambda g: (sum_to(g / b, sa), sum_to(neg(g * a) / (b * b), sb)))


def pow(a, b):  # noqa: A001
    if isinstance(b, Variable):
        if isinstance(a, Variable):
            raise NotImplementedError("Variable ** Variable")
        # constant ** Variable
        out = _raw(a) ** b.array
        return make(out, (b,), lambda g: (g * Variable(out) * np.log(a),))
    return make(_raw(a) ** b, (a,), lambda g: (g * b * pow(a, b - 1),))


def neg(a):
    return make(-_raw(a), (a,), lambda g: (neg(g),))


def absolute(a):
    return make(abs(_raw(a)), (a,), lambda g: (g * np.sign(_raw(a)),))


def swapaxes(x, a1, a2):
    return make(np.swapaxes(_raw(x), a1, a2), (x,), lambda g: (swapaxes(g, a1, a2),))


def np_matmul(a, b):
    """numpy `@` semantics (the reference uses it on 1-D, 2-D and batched operands)"""
    ra, rb = _raw(a), _raw(b)
    out = np.matmul(ra, rb)
    sa, sb = ra.shape, rb.shape

    def vjp(g):
        if len(sa) == 1 and len(sb) == 1:
            return g * b, g * a
        if len(sa) == 1:            # (n) @ (..., n, m) -> (..., m)
            ga = sum_to(np_matmul(b, expand_dims(g, -1)), sb[:-2] + (sa[0], 1))
            ga = reshape(sum(ga, axis=tuple(range(len(sb) - 2))) if len(sb) > 2 else ga, sa)
            gb = expand_dims(a, -1) * expand_dims(g, -2)
            return ga, sum_to(gb, sb)
        if len(sb) == 1:            # (..., n, m) @ (m) -> (..., n)
            ga = expand_dims(g, -1) * b
            gb = np_matmul(swapaxes(a, -1, -2), expand_dims(g, -1))
            gb = reshape(sum(gb, axis=tuple(range(len(sa) - 2))) if len(sa) > 2 else gb, sb)
            return sum_to(ga, sa), gb
        ga = np_matmul(g, swapaxes(b, -1, -2))
        gb = np_matmul(swapaxes(a, -1, -2), g)
        return sum_to(ga, sa), sum_to(gb, sb)

    return make(out, (a, b), vjp)


def matmul(a, b, transa=False, transb=False):
    if transa:
        a = swapaxes(a, -1, -2)
    if transb:
        b = swapaxes(b, -1, -2)
    return np_matmul(a, b)


def transpose(x, axes=None):
    if axes is None:
        inv = None
    else:
        inv = tuple(int(i) for i in np.argsort(axes))
    return make(np.transpose(_raw(x), axes), (x,), lambda g: (transpose(g, inv),))


def batch_inv(a):
    out = np.linalg.inv(_raw(a))

    def vjp(g):
        o = Variable(out)
        ot = swapaxes(o, -1, -2)
        return (neg(np_matmul(np_matmul(ot, g), ot)),)

    return make(out, (a,), vjp)


def reshape(x, shape):
    xs = _shape(x)
    return make(np.reshape(_raw(x), shape), (x,), lambda g: (reshape(g, xs),))


def squeeze(x, axis=None):
    xs = _shape(x)
    return make(np.squeeze(_raw(x), axis=axis), (x,), lambda g: (reshape(g, xs),))


def expand_dims(x, axis):
    xs = _shape(x)
    return make(np.expand_dims(_raw(x), axis), (x,), lambda g: (reshape(g, xs),))


class _Scatter:
    pass


def _scatter(g, idx, shape):
    """zeros(shape)[idx] += g  (the vector-Jacobian product of get_item)"""
    out = np.zeros(shape, dtype=_raw(g).dtype)
    np.add.at(out, idx, _raw(g))
    return make(out, (g,), lambda gg: (get_item(gg, idx),))


def get_item(x, idx):
    if isinstance(idx, tuple):
        idx = tuple(_raw(i) for i in idx)
    else:
        idx = _raw(idx)
    xs = _shape(x)
    return make(_raw(x)[idx], (x,), lambda g: (_scatter(g, idx, xs),))


def _axis_index(ndim, axis, sl):
    axis = axis % ndim
    return (slice(None),) * axis + (sl,)


def concat(xs, axis=1):
    xs = list(xs)
    out = np.concatenate([_raw(x) for x in xs], axis=axis)
    sizes = [_shape(x)[axis] for x in xs]
    offs = np.cumsum([0] + sizes)

    def vjp(g):
        return tuple(get_item(g, _axis_index(out.ndim, axis, slice(int(offs[i]), int(offs[i + 1]))))
                     for i in range(len(xs)))

    return make(out, tuple(xs), vjp)


def stack(xs, axis=0):
    xs = list(xs)
    out = np.stack([_raw(x) for x in xs], axis=axis)

    def vjp(g):
        return tuple(get_item(g, _axis_index(out.ndim, axis, i)) for i in range(len(xs)))

    return make(out, tuple(xs), vjp)


def where(cond, x, y):
    c = _raw(cond)
    sx, sy = _shape(x), _shape(y)

    def vjp(g):
        z = np.zeros((), dtype=_raw(g).dtype)
        return None, sum_to(where(c, g, z), sx), sum_to(where(c, z, g), sy)

    return make(np.where(c, _raw(x), _raw(y)), (None, x, y), vjp)


def cast(x, typ):
    src = _raw(x).dtype
    return make(_raw(x).astype(typ), (x,), lambda g: (cast(g, src),))


def repeat(x, repeats, axis=None):
    xs = _shape(x)
    out = np.repeat(_raw(x), repeats, axis=axis)

    def vjp(g):
        if axis is None:
            return (reshape(sum(reshape(g, (-1, repeats)), axis=1), xs),)
        ax = axis % len(xs)
        folded = reshape(g, xs[:ax] + (xs[ax], repeats) + xs[ax + 1:])
        return (sum(folded, axis=ax + 1),)

    return make(out, (x,), vjp)


def minimum(a, b):
    ra, rb = _raw(a), _raw(b)
    sa, sb = ra.shape, rb.shape
    m = ra <= rb
    return make(np.minimum(ra, rb), (a, b), lambda g: (sum_to(g * m, sa), sum_to(g * (~m), sb)))


def maximum(a, b):
    ra, rb = _raw(a), _raw(b)
    sa, sb = ra.shape, rb.shape
    m = ra >= rb
    return make(np.maximum(ra, rb), (a, b), lambda g: (sum_to(g * m, sa), sum_to(g * (~m), sb)))


def sum(x, axis=None, keepdims=False):  # noqa: A001
    xs = _shape(x)
    out = np.sum(_raw(x), axis=axis, keepdims=keepdims)

    def vjp(g):
        if axis is None:
            kshape = (1,) * len(xs)
        else:
            ax = (axis,) if isinstance(axis, int) else tuple(axis)
            ax = tuple(a % len(xs) for a in ax)
            kshape = tuple(1 if i in ax else s for i, s in enumerate(xs))
        return (broadcast_to(reshape(g, kshape), xs),)

    return make(out, (x,), vjp)


def mean(x, axis=None, keepdims=False):
    xs = _shape(x)
    out = np.mean(_raw(x), axis=axis, keepdims=keepdims)
    n = (np.prod(xs) if xs else 1) / max(1, out.size)

    def vjp(g):
        if axis is None:
            kshape = (1,) * len(xs)
        else:
            ax = (axis,) if isinstance(axis, int) else tuple(axis)
            ax = tuple(a % len(xs) for a in ax)
            kshape = tuple(1 if i in ax else s for i, s in enumerate(xs))
        return (broadcast_to(reshape(g, kshape), xs) / float(n),)

    return make(out, (x,), vjp)


def split_axis(x, indices_or_sections, axis, force_tuple=True):
    parts = np.split(_raw(x), indices_or_sections, axis=axis)
    nd = _raw(x).ndim
    res, off = [], 0
    for p in parts:
        n = p.shape[axis]
        res.append(get_item(x, _axis_index(nd, axis, slice(off, off + n))))
        off += n
    return tuple(res)


def separate(x, axis=0):
    nd = _raw(x).ndim
    return tuple(get_item(x, _axis_index(nd, axis, i)) for i in range(_shape(x)[axis]))


def sigmoid(x):
    out = 1.0 / (1.0 + np.exp(-_raw(x)))

    def vjp(g):
        y = sigmoid(x)
        return (g * y * (1.0 - y),)

    return make(out, (x,), vjp)


def sqrt(x):
    return make(np.sqrt(_raw(x)), (x,), lambda g: (g / (2.0 * sqrt(x)),))


def exp(x):
    return make(np.exp(_raw(x)), (x,), lambda g: (g * exp(x),))


def log(x):
    return make(np.log(_raw(x)), (x,), lambda g: (g / x,))


def mean_squared_error(a, b):
    d = a - b if isinstance(a, Variable) or isinstance(b, Variable) else Variable(_raw(a) - _raw(b))
    return mean(d * d)


# ---- what env_dx/pendulum.py:65-102 (PendulumDx.forward) needs on top of the hot path's list
# Derivative of F.clip AT the limits.  Chainer's ClipGrad is TAKEN to be inclusive, (x_min <= x) & (x <= x_max); Chainer is
# not installable here, so this is an assumption and not a pin.  Every fixture that goes through
# chainer.grad of PendulumDx.forward (pendulum.npz, pendulum_boxddp.npz, imitation_*.npz) inherits the setting; the
# product's counterpart is PendulumDx.clamp_grad_closed.
CLIP_GRAD_CLOSED = True


def clip(x, x_min, x_max):
    """F.clip; derivative 1 inside (x_min, x_max), CLIP_GRAD_CLOSED at the two limits, 0 outside"""
    r = _raw(x)
    inside = ((x_min <= r) & (r <= x_max)) if CLIP_GRAD_CLOSED else ((x_min < r) & (r < x_max))
    return make(np.clip(r, x_min, x_max), (x,), lambda g: (g * inside,))


def arctan2(a, b):
    sa, sb = _shape(a), _shape(b)

    def vjp(g):
        r2 = a * a + b * b
        return sum_to(g * b / r2, sa), sum_to(neg(g * a) / r2, sb)

    return make(np.arctan2(_raw(a), _raw(b)), (a, b), vjp)


def sin(x):
    return make(np.sin(_raw(x)), (x,), lambda g: (g * cos(x),))


def cos(x):
    return make(np.cos(_raw(x)), (x,), lambda g: (neg(g * sin(x)),))


def scatter_add(a, slices, b):
    out = np.array(_raw(a), copy=True)
    np.add.at(out, slices, _raw(b))
    return make(out, (a, b), lambda g: (g, get_item(g, slices)))
