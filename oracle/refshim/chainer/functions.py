"""`chainer.functions` stand-in (only what the reference's hot path calls)."""
import numpy as np

from .variable import Variable, _raw


def _v(x):
    return Variable(x)


def matmul(a, b, transa=False, transb=False):
    a, b = _raw(a), _raw(b)
    if transa:
        a = np.swapaxes(a, -1, -2)
    if transb:
        b = np.swapaxes(b, -1, -2)
    return _v(np.matmul(a, b))


def transpose(x, axes=None):
    return _v(np.transpose(_raw(x), axes))


def batch_inv(a):
    return _v(np.linalg.inv(_raw(a)))


def squeeze(x, axis=None):
    return _v(np.squeeze(_raw(x), axis=axis))


def expand_dims(x, axis):
    return _v(np.expand_dims(_raw(x), axis))


def concat(xs, axis=1):
    return _v(np.concatenate([_raw(x) for x in xs], axis=axis))


def stack(xs, axis=0):
    return _v(np.stack([_raw(x) for x in xs], axis=axis))


def where(cond, x, y):
    return _v(np.where(_raw(cond), _raw(x), _raw(y)))


def cast(x, typ):
    return _v(_raw(x).astype(typ))


def repeat(x, repeats, axis=None):
    return _v(np.repeat(_raw(x), repeats, axis=axis))


def minimum(a, b):
    return _v(np.minimum(_raw(a), _raw(b)))


def maximum(a, b):
    return _v(np.maximum(_raw(a), _raw(b)))


def sum(x, axis=None, keepdims=False):  # noqa: A001
    return _v(np.sum(_raw(x), axis=axis, keepdims=keepdims))


def mean(x, axis=None, keepdims=False):
    return _v(np.mean(_raw(x), axis=axis, keepdims=keepdims))


def split_axis(x, indices_or_sections, axis, force_tuple=True):
    return tuple(_v(p) for p in np.split(_raw(x), indices_or_sections, axis=axis))


def sigmoid(x):
    return _v(1.0 / (1.0 + np.exp(-_raw(x))))


def sqrt(x):
    return _v(np.sqrt(_raw(x)))


def reshape(x, shape):
    return _v(np.reshape(_raw(x), shape))


def mean_squared_error(a, b):
    d = _raw(a) - _raw(b)
    return _v(np.mean(d * d))
