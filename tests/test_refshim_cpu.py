"""CPU: the numpy stand-in for `chainer` that lets tests/golden/make_golden.py run the unmodified reference
(oracle/refshim/chainer - test infrastructure) carries a small reverse-mode tape.  Its derivatives are checked here
against central differences, operation by operation, incl. a second derivative (the reference's approximate_cost
uses chainer.grad with enable_double_backprop, mpc/approximate.py:36-45)."""
import importlib
import os
import sys

import numpy as np
import pytest

SHIM = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "refshim")


@pytest.fixture(scope="module")
def ch():
    saved = sys.modules.pop("chainer", None)
    sys.path.insert(0, SHIM)
    try:
        mod = importlib.import_module("chainer")
        assert mod.__file__.startswith(SHIM)
        yield mod
    finally:
        sys.path.remove(SHIM)
        for k in [k for k in sys.modules if k == "chainer" or k.startswith("chainer.")]:
            del sys.modules[k]
        if saved is not None:
            sys.modules["chainer"] = saved


def fd(f, x, eps=1e-6):
    g = np.zeros_like(x)
    for i in np.ndindex(x.shape):
        xp, xm = x.copy(), x.copy()
        xp[i] += eps
        xm[i] -= eps
        g[i] = (f(xp) - f(xm)) / (2 * eps)
    return g


def test_tape_gradients_match_central_differences(ch):
    F = ch.functions
    rng = np.random.RandomState(0)
    cases = [
        ("add/mul broadcast", lambda a, b: a + b * 2.0, [(3, 4), (4,)]),
        ("div", lambda a, b: a / (b * b + 1.0), [(3, 4), (3, 1)]),
        ("matmul batched", lambda a, b: a @ b, [(2, 3, 4), (2, 4, 5)]),
        ("matmul 1d @ 2d", lambda a, b: a @ b, [(4,), (4, 5)]),
        ("matmul 2d @ 1d", lambda a, b: a @ b, [(3, 4), (4,)]),
        ("F.matmul transa", lambda a, b: F.matmul(a, b, transa=True), [(2, 4, 3), (2, 4, 5)]),
        ("batch_inv", lambda a: F.batch_inv(a + 3 * np.eye(3)), [(2, 3, 3)]),
        ("get_item", lambda a: a[:, 1:3] * a[:, 0:2], [(3, 4)]),
        ("concat", lambda a, b: F.concat((a, b), axis=1) ** 2, [(3, 2), (3, 3)]),
        ("stack", lambda a, b: F.stack((a, b), axis=1) ** 3, [(3, 2), (3, 2)]),
        ("where", lambda a, b: F.where(np.eye(3, dtype=bool), a, b * b), [(3,), (3, 3)]),
        ("repeat/expand_dims", lambda a: F.repeat(F.expand_dims(a, 0), 4, axis=0) * np.arange(12.).reshape(4, 3), [(3,)]),
        ("minimum/maximum", lambda a, b: F.minimum(F.maximum(a, b), b + 0.5), [(5,), (5,)]),
        ("sigmoid/sqrt", lambda a: F.sqrt(F.sigmoid(a)) * a, [(4,)]),
        ("arctan2/sin/cos", lambda a, b: F.sin(F.arctan2(a, b)) + F.cos(a * b), [(4,), (4,)]),
        ("transpose", lambda a: F.transpose(a, (1, 0, 2)) * np.arange(24.).reshape(3, 2, 4), [(2, 3, 4)]),
        ("mean", lambda a: F.mean(a, axis=1) * np.arange(3.), [(3, 4)]),
        ("separate", lambda a: F.separate(a, axis=1)[0] * F.separate(a, axis=1)[2], [(3, 3)]),
        ("split_axis", lambda a: F.split_axis(a, 3, axis=1)[1] ** 2, [(3, 3)]),
        ("mean_squared_error", lambda a, b: F.mean_squared_error(a, b), [(3, 3), (3, 3)]),
        ("clip", lambda a: F.clip(a * 3, -2., 2.) ** 2, [(6,)]),
    ]
    for name, fn, shapes in cases:
        xs = [rng.randn(*s) + 0.1 for s in shapes]
        vs = [ch.Variable(x) for x in xs]
        gs = ch.grad([F.sum(fn(*vs))], vs)
        for k, (x, g) in enumerate(zip(xs, gs)):
            def f(xx, k=k):
                args = list(xs)
                args[k] = xx
                return float(np.sum(fn(*[ch.Variable(t) for t in args]).array))
            assert np.abs(fd(f, x) - g.array).max() < 1e-5, (name, k)


def test_tape_second_derivative_and_function_node(ch):
    F = ch.functions
    x = ch.Variable(np.array([0.3, -1.2, 2.0, 0.7]))
    g = ch.grad([F.sum(x ** 3)], [x], enable_double_backprop=True)[0]
    h = ch.grad([F.sum(g[1:2])], [x])[0]                      # row 1 of the Hessian of sum x^3: 6 x_1 e_1
    np.testing.assert_allclose(h.array, [0.0, 6 * -1.2, 0.0, 0.0], atol=1e-12)
    # d clip / dx is 1 ON the bounds (what the pendulum fixture's 0.15 at u = +-2 rests on)
    u = ch.Variable(np.array([-2.5, -2.0, 0.0, 2.0, 2.5]))
    gu = ch.grad([F.sum(F.clip(u, -2.0, 2.0))], [u])[0]
    np.testing.assert_array_equal(gu.array, [0.0, 1.0, 1.0, 1.0, 0.0])
    # no_backprop_mode records nothing
    with ch.no_backprop_mode():
        y = x * x
    assert y.creator is None
