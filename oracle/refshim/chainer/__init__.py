"""numpy stand-in for the `chainer` package (array wrapper + a small reverse-mode tape).

TEST INFRASTRUCTURE ONLY.  Chainer is not installable in the build container (no
network), yet every hot-path module of the reference imports it at module scope.
The reference's gradients on this path are hand-written (`DiffLqr.backward`,
`MPCstep.backward`), so Chainer is used only as a thin array wrapper there.  This
package provides exactly that wrapper so that `tests/golden/make_golden.py` can
execute the *unmodified* reference sources from /root/reference and record golden
input/output vectors.  It is never imported by the product package, by bench.py's
GPU path, or on the GPU box (the reference does not travel there).

`chainer.grad` / `Variable.backward` are a small tape over numpy with exact analytic derivatives (variable.py,
functions.py) - enough for mpc/approximate.py (`linearize_dynamics`, `approximate_cost` with double backprop),
BoxDDP around the non-linear PendulumDx and the imitation step's gradient through `MPCstep.backward`.
Not provided: optimizers, iterators, serializers.
"""
import contextlib

import numpy as _np

from . import backend  # noqa: F401
from . import function_node  # noqa: F401
from . import functions  # noqa: F401
from . import utils  # noqa: F401
from . import variable as _variable
from .variable import Parameter, Variable, as_variable, grad  # noqa: F401

_variable._install_operators()

__version__ = "0.0-shim"


class Link:
    """Minimal `chainer.Link`: parameter scope + `__call__ -> forward`."""

    xp = _np

    def __init__(self):
        pass

    @contextlib.contextmanager
    def init_scope(self):
        yield

    def __call__(self, *args, **kwargs):
        return self.forward(*args, **kwargs)

    def params(self):
        for v in self.__dict__.values():
            if isinstance(v, Parameter):
                yield v

    def cleargrads(self):
        for p in self.params():
            p.grad_var = None


Chain = Link


@contextlib.contextmanager
def no_backprop_mode():
    prev = _variable._state["record"]
    _variable._state["record"] = False
    try:
        yield
    finally:
        _variable._state["record"] = prev
