"""Import the UNMODIFIED reference (pfnet-research/chainer-differentiable-mpc) in the
build container, on top of the numpy `chainer` stand-in that lives next to this file.

TEST INFRASTRUCTURE ONLY - used by tests/golden/make_golden.py (run here, where
/root/reference exists) to record golden vectors.  Nothing on the GPU box imports
this: /root/reference does not exist there.

Two compatibility fixes are applied *outside* the reference tree:
  1. `chainer` resolves to oracle/refshim/chainer (forward-only numpy wrapper).
  2. `torch.lu_solve(b, LU, piv)` with a 2-D `b` and 3-D `LU` is given back its
     torch<=1.2 (`btrisolve`) meaning - the reference's `xpbatch_lu_solve`
     (util.py:505-528) calls it that way (pnqp.py:83,137) and torch 2.x rejects it.
     `torch.lu` is mapped to `torch.linalg.lu_factor` (same LAPACK getrf, same
     1-based int32 pivots) where the deprecated alias is missing.
"""
import importlib
import os
import sys

REFERENCE_ROOT = os.environ.get("DMPC_REFERENCE_ROOT", "/root/reference")
_SHIM_DIR = os.path.dirname(os.path.abspath(__file__))

_loaded = None


def _patch_torch():
    import torch

    if getattr(torch, "_dmpc_ref_patched", False):
        return
    _orig_lu_solve = torch.lu_solve

    def lu_solve_compat(b, LU, pivots):
        if b.dim() == LU.dim() - 1:
            return _orig_lu_solve(b.unsqueeze(-1), LU, pivots).squeeze(-1)
        return _orig_lu_solve(b, LU, pivots)

    torch.lu_solve = lu_solve_compat
    if not hasattr(torch, "lu"):
        def lu_compat(A, pivot=True, get_infos=False):
            LU, piv = torch.linalg.lu_factor(A, pivot=pivot)
            return LU, piv
        torch.lu = lu_compat
    torch._dmpc_ref_patched = True


def available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "lqr"))


def load():
    """Return a namespace with the reference's hot-path modules."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not found at %s" % REFERENCE_ROOT)
    for p in (os.path.join(REFERENCE_ROOT, "env_dx"), os.path.join(REFERENCE_ROOT, "mpc"),
              os.path.join(REFERENCE_ROOT, "lqr"), REFERENCE_ROOT, _SHIM_DIR):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)
    for name in ("chainer", "util", "lqr_recursion", "differentiable_lqr", "pnqp",
                 "mpc_step", "active_constrained_lqr", "box_ddp", "approximate", "pendulum", "il_env"):
        if name in sys.modules and not getattr(sys.modules[name], "__file__", "").startswith(
                (REFERENCE_ROOT, _SHIM_DIR)):
            del sys.modules[name]
    _patch_torch()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ns = type("ReferenceModules", (), {})()
        ns.chainer = importlib.import_module("chainer")
        ns.util = importlib.import_module("util")
        ns.lqr_recursion = importlib.import_module("lqr_recursion")
        ns.differentiable_lqr = importlib.import_module("differentiable_lqr")
        ns.pnqp = importlib.import_module("pnqp")
        ns.active_constrained_lqr = importlib.import_module("active_constrained_lqr")
        ns.mpc_step = importlib.import_module("mpc_step")
        try:
            ns.box_ddp = importlib.import_module("box_ddp")
            ns.approximate = importlib.import_module("approximate")
        except Exception as e:
            ns.box_ddp = ns.approximate = None
            ns.box_ddp_error = e
        # env_dx/pendulum.py (PendulumDx forward model, true objective) and env_dx/il_env.py (sample_xinit, mpc):
        # pendulum.py imports matplotlib at module scope (Agg backend, present in this image)
        try:
            ns.pendulum = importlib.import_module("pendulum")
            ns.il_env = importlib.import_module("il_env")
        except Exception as e:
            ns.pendulum = ns.il_env = None
            ns.env_dx_error = e
    _loaded = ns
    return ns
