"""`chainer.Variable` stand-in: an ndarray holder with operator forwarding."""
import numpy as np


def _raw(x):
    return x.array if isinstance(x, Variable) else x


class Variable:
    # make `ndarray <op> Variable` dispatch to Variable.__r<op>__
    __array_priority__ = 200

    def __init__(self, data=None, name=None, requires_grad=True):
        if isinstance(data, Variable):
            data = data.array
        self.array = None if data is None else np.asarray(data)
        self.grad = None
        self.name = name

    # --- array-ish surface -------------------------------------------------
    @property
    def data(self):
        return self.array

    @data.setter
    def data(self, v):
        self.array = v

    @property
    def shape(self):
        return self.array.shape

    @property
    def dtype(self):
        return self.array.dtype

    @property
    def ndim(self):
        return self.array.ndim

    @property
    def size(self):
        return self.array.size

    @property
    def T(self):
        return Variable(self.array.T)

    def __len__(self):
        return len(self.array)

    def __getitem__(self, idx):
        if isinstance(idx, tuple):
            idx = tuple(_raw(i) for i in idx)
        else:
            idx = _raw(idx)
        return Variable(self.array[idx])

    def reshape(self, *shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return Variable(self.array.reshape(shape))

    def __repr__(self):
        return "variable(%r)" % (self.array,)

    def __array__(self, dtype=None, copy=None):
        a = self.array
        return a if dtype is None else a.astype(dtype)

    def __float__(self):
        return float(self.array)

    def __bool__(self):
        return bool(self.array)

    def __format__(self, spec):
        return format(self.array.item() if self.array.ndim == 0 else self.array, spec)

    def __deepcopy__(self, memo):
        return Variable(np.array(self.array, copy=True))

    def backward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("forward-only stand-in")

    def cleargrad(self):
        self.grad = None

    def unchain_backward(self):
        pass

    # --- arithmetic ----------------------------------------------------------
    def __neg__(self):
        return Variable(-self.array)

    def __pos__(self):
        return self

    def __abs__(self):
        return Variable(abs(self.array))

    # comparisons return raw ndarrays (what the reference relies on)
    def __lt__(self, o):
        return self.array < _raw(o)

    def __le__(self, o):
        return self.array <= _raw(o)

    def __gt__(self, o):
        return self.array > _raw(o)

    def __ge__(self, o):
        return self.array >= _raw(o)

    def __eq__(self, o):
        return self.array == _raw(o)

    def __ne__(self, o):
        return self.array != _raw(o)

    __hash__ = object.__hash__


def _binop(name, fn):
    def fwd(self, other):
        return Variable(fn(self.array, _raw(other)))

    def rev(self, other):
        return Variable(fn(_raw(other), self.array))

    setattr(Variable, "__%s__" % name, fwd)
    setattr(Variable, "__r%s__" % name, rev)
    setattr(Variable, "__i%s__" % name, fwd)


_binop("add", lambda a, b: a + b)
_binop("sub", lambda a, b: a - b)
_binop("mul", lambda a, b: a * b)
_binop("truediv", lambda a, b: a / b)
_binop("pow", lambda a, b: a ** b)
_binop("matmul", lambda a, b: np.matmul(a, b))


class Parameter(Variable):
    def __init__(self, initializer=None, shape=None, name=None):
        super().__init__(initializer, name=name)


def as_variable(x):
    return x if isinstance(x, Variable) else Variable(x)
