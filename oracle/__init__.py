"""CPU oracle for the differentiable-MPC hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It is a plain-numpy restatement (float64, op-for-op, quirks included) of the
reference's algorithm for the hot path named by BASELINE.json `north_star`
(pfnet-research/chainer-differentiable-mpc):

    oracle/linalg.py   <- util.py:101-123,280-358,411-434,462-528
    oracle/lqr.py      <- lqr/lqr_recursion.py:69-209
    oracle/kkt.py      <- lqr/differentiable_lqr.py:78-142
    oracle/pnqp.py     <- mpc/pnqp.py:26-201
    oracle/mpc.py      <- mpc/mpc_step.py:70-460, mpc/active_constrained_lqr.py:67-202,
                          util.py:162-198
    oracle/refshim/    <- a numpy stand-in for `chainer` + a loader that runs the
                          UNMODIFIED reference from /root/reference in the build
                          container (golden-vector generation only).

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py`
may import it - as the checker / reported baseline, never as the thing shipped or
measured.  The product package `chainer_differentiable_mpc_amd` never imports it
and fails loudly when its HIP library is missing.

Parity pinning: the reference has no automated tests.  The oracle is pinned by
  (1) the notebook known answers the reference holds:
      one-variable LQR gains/states, Boyd LQR gains, the PNQP 2x4 answer, the
      LQRnet iteration-0 loss;
  (2) golden vectors produced by running the unmodified reference here through
      `oracle/refshim` (tests/golden/*.npz + tests/golden/make_golden.py).
Both are checked by `tests/test_oracle_golden.py` (CPU, `-m "not gpu"`).
"""
