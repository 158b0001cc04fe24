"""`chainer.Variable` stand-in: an ndarray holder with operator forwarding and a small reverse-mode tape.

TEST INFRASTRUCTURE ONLY (see __init__.py).  Forward values are plain numpy (unchanged arithmetic);
every differentiable operation additionally records a node (inputs, vector-Jacobian product).  The
vector-Jacobian products are themselves written with Variable operations, so a gradient can be
differentiated again - `chainer.grad(..., enable_double_backprop=True)` as used by the reference's
mpc/approximate.py:18-54.  Derivatives are the exact analytic ones, i.e. what Chainer's own backward
implementations compute up to float64 rounding.
"""
import heapq
import itertools

import numpy as np

_counter = itertools.count()
_state = {"record": True}


def _raw(x):
    return x.array if isinstance(x, Variable) else x


class _Node:
    """creator of one or more Variables: `vjp(list of output grads) -> list of input grads` (Variables or None)"""
    __slots__ = ("inputs", "vjp", "rank", "outputs")

    def __init__(self, inputs, vjp):
        self.inputs = inputs
        self.vjp = vjp
        self.rank = next(_counter)
        self.outputs = []


def recording():
    return _state["record"]


def make(value, inputs=(), vjp=None):
    """wrap `value` as the single output of an operation on `inputs` (raw arrays are constants)"""
    out = Variable(value)
    if vjp is not None and _state["record"] and any(isinstance(i, Variable) for i in inputs):
        node = _Node(list(inputs), lambda gys: vjp(gys[0]))
        node.outputs.append(out)
        out.creator = node
    return out


def make_multi(values, inputs, vjp):
    outs = [Variable(v) for v in values]
    if vjp is not None and _state["record"] and any(isinstance(i, Variable) for i in inputs):
        node = _Node(list(inputs), vjp)
        for i, o in enumerate(outs):
            o.creator = node
            o.out_index = i
            node.outputs.append(o)
    return outs


class Variable:
    # make `ndarray <op> Variable` dispatch to Variable.__r<op>__
    __array_priority__ = 200

    def __init__(self, data=None, name=None, requires_grad=True):
        if isinstance(data, Variable):
            data = data.array
        self.array = None if data is None else np.asarray(data)
        self.grad_var = None
        self.name = name
        self.creator = None
        self.out_index = 0

    # --- array-ish surface -------------------------------------------------
    @property
    def data(self):
        return self.array

    @data.setter
    def data(self, v):
        self.array = v

    @property
    def grad(self):
        return None if self.grad_var is None else self.grad_var.array

    @grad.setter
    def grad(self, g):
        self.grad_var = None if g is None else Variable(g)

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
        from . import functions as F
        return F.transpose(self)

    def __len__(self):
        return len(self.array)

    def __getitem__(self, idx):
        from . import functions as F
        return F.get_item(self, idx)

    def reshape(self, *shape):
        from . import functions as F
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        return F.reshape(self, shape)

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

    def backward(self, retain_grad=False, enable_double_backprop=False, loss_scale=None):
        """accumulate d self / d leaf into `leaf.grad` of every leaf Variable reached"""
        grads, seen = _backprop([self], None, enable_double_backprop)
        for v in seen.values():
            if v.creator is None and id(v) in grads and v is not self:
                g = grads[id(v)]
                v.grad_var = g if v.grad_var is None else Variable(v.grad_var.array + g.array)

    def cleargrad(self):
        self.grad_var = None

    def unchain_backward(self):
        self.creator = None

    def unchain(self):
        self.creator = None

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


def _backprop(outputs, grad_outputs, double):
    """reverse sweep from `outputs`; returns ({id(var): grad Variable}, {id(var): var})"""
    grads, seen = {}, {}
    heap, queued = [], set()

    def push(v):
        seen[id(v)] = v
        n = v.creator
        if n is not None and id(n) not in queued:
            queued.add(id(n))
            heapq.heappush(heap, (-n.rank, id(n), n))

    for i, y in enumerate(outputs):
        g = None if grad_outputs is None else grad_outputs[i]
        g = Variable(np.ones_like(y.array)) if g is None else (g if isinstance(g, Variable) else Variable(g))
        grads[id(y)] = g if id(y) not in grads else _acc(grads[id(y)], g)
        push(y)
    prev = _state["record"]
    _state["record"] = bool(double) and prev
    try:
        while heap:
            _, _, n = heapq.heappop(heap)
            gys = [grads.get(id(o)) for o in n.outputs]
            if all(g is None for g in gys):
                continue
            gxs = n.vjp(gys)
            for x, gx in zip(n.inputs, gxs):
                if gx is None or not isinstance(x, Variable):
                    continue
                if not isinstance(gx, Variable):
                    gx = Variable(gx)
                grads[id(x)] = gx if id(x) not in grads else _acc(grads[id(x)], gx)
                push(x)
    finally:
        _state["record"] = prev
    return grads, seen


def _acc(a, b):
    return a + b


def grad(outputs, inputs, grad_outputs=None, grad_inputs=None, set_grad=False, retain_grad=False,
         enable_double_backprop=False, loss_scale=None):
    """`chainer.grad`: d sum(outputs) / d inputs as a list of Variables (None where unreachable)"""
    grads, _ = _backprop(list(outputs), grad_outputs, enable_double_backprop)
    res = [grads.get(id(x)) for x in inputs]
    if set_grad:
        for x, g in zip(inputs, res):
            x.grad_var = g
    return res


def _install_operators():
    from . import functions as F

    def binop(name, fn):
        def fwd(self, other):
            return fn(self, other)

        def rev(self, other):
            return fn(other, self)

        setattr(Variable, "__%s__" % name, fwd)
        setattr(Variable, "__r%s__" % name, rev)
        setattr(Variable, "__i%s__" % name, fwd)

    binop("add", F.add)
    binop("sub", F.sub)
    binop("mul", F.mul)
    binop("truediv", F.div)
    binop("pow", F.pow)
    binop("matmul", F.np_matmul)
    Variable.__neg__ = lambda self: F.neg(self)
    Variable.__pos__ = lambda self: self
    Variable.__abs__ = lambda self: F.absolute(self)


class Parameter(Variable):
    def __init__(self, initializer=None, shape=None, name=None):
        super().__init__(initializer, name=name)
        self.update_rule = type("UpdateRule", (), {"enabled": True})()


def as_variable(x):
    return x if isinstance(x, Variable) else Variable(x)
