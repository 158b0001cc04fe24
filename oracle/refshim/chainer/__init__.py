"""Forward-only numpy stand-in for the `chainer` package.

TEST INFRASTRUCTURE ONLY.  Chainer is not installable in the build container (no
network), yet every hot-path module of the reference imports it at module scope.
The reference's gradients on this path are hand-written (`DiffLqr.backward`,
`MPCstep.backward`), so Chainer is used only as a thin array wrapper there.  This
package provides exactly that wrapper so that `tests/golden/make_golden.py` can
execute the *unmodified* reference sources from /root/reference and record golden
input/output vectors.  It is never imported by the product package, by bench.py's
GPU path, or on the GPU box (the reference does not travel there).

Not provided: autograd (`chainer.grad`), optimizers, iterators, serializers.
"""
import contextlib

import numpy as _np

from . import backend  # noqa: F401
from . import function_node  # noqa: F401
from . import functions  # noqa: F401
from . import utils  # noqa: F401
from .variable import Parameter, Variable, as_variable  # noqa: F401

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
            p.grad = None


Chain = Link


@contextlib.contextmanager
def no_backprop_mode():
    yield


def grad(*args, **kwargs):  # pragma: no cover - documented gap
    raise NotImplementedError("the numpy stand-in has no autograd (chainer.grad)")
