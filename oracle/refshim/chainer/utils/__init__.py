"""`chainer.utils` stand-in."""
from . import type_check  # noqa: F401
