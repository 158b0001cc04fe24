"""`chainer.backend` stand-in: everything is numpy."""
import numpy as _np


def get_array_module(*args):
    return _np
