"""ctypes binding of libdmpc_hip.so (the C-ABI in include/dmpc.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every solver call below is
one C function taking raw device pointers.  There is NO CPU fallback: if the shared library is
missing or no GPU is visible the product path raises.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libdmpc_hip.so"
LIB_PATH = os.environ.get("DMPC_LIB", os.path.join(_HERE, LIB_NAME))

ABI_VERSION = 411           # include/dmpc.h: DMPC_VERSION the signatures below were written for
E_BADARG, E_UNSUPPORTED, E_WORKSPACE = -1, -2, -3
INFO_SINGULAR, INFO_NONFINITE, INFO_QP_ITERCAP, INFO_LS_ITERCAP = 1, 2, 4, 8

_c_f = ctypes.c_void_p      # const float* / float* (device)
_c_i = ctypes.c_int
_c_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/dmpc.h one to one
SIGNATURES = {
    "dmpc_version": (_c_i, []),
    "dmpc_source_hash": (ctypes.c_char_p, []),
    "dmpc_last_kernel_name": (_c_i, [ctypes.c_char_p, _c_sz]),
    "dmpc_lqr_kernel_family": (_c_i, [_c_i, _c_i]),
    "dmpc_lqr_solve_path": (_c_i, [_c_i] * 4),
    "dmpc_lqr_workspace_bytes": (_c_sz, [_c_i] * 4),
    "dmpc_lqr_solve": (_c_i, [_c_i] * 4 + [_c_f] * 10 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_lqr_saving_available": (_c_i, [_c_i] * 4),
    "dmpc_lqr_solve_saving": (_c_i, [_c_i] * 4 + [_c_f] * 14),
    "dmpc_lqr_saved_solve": (_c_i, [_c_i] * 4 + [_c_f] * 10),
    "dmpc_lqr_backward_sweep": (_c_i, [_c_i] * 4 + [_c_f] * 9),
    "dmpc_lqr_backward_sweep_ws": (_c_i, [_c_i] * 4 + [_c_f] * 7 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_lqr_forward_sweep": (_c_i, [_c_i] * 4 + [_c_f] * 10),
    "dmpc_lqr_kkt_workspace_bytes": (_c_sz, [_c_i] * 4),
    "dmpc_lqr_kkt_grad": (_c_i, [_c_i] * 4 + [_c_f] * 7 + [_c_i] + [_c_f] * 5 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_lqr_kkt_grad_saved": (_c_i, [_c_i] * 4 + [_c_f] * 11 + [_c_i] + [_c_f] * 5 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_lqr_f64_path": (_c_i, [_c_i, _c_i]),
    "dmpc_lqr_f64_workspace_bytes": (_c_sz, [_c_i] * 4),
    "dmpc_lqr_solve_f64": (_c_i, [_c_i] * 4 + [_c_f] * 10 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_lqr_kkt_grad_f64": (_c_i, [_c_i] * 4 + [_c_f] * 7 + [_c_i] + [_c_f] * 5 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_batch_lu_factor": (_c_i, [_c_i, _c_i] + [_c_f] * 5),
    "dmpc_batch_lu_solve": (_c_i, [_c_i, _c_i, _c_i] + [_c_f] * 5),
    "dmpc_coupled_workspace_bytes": (_c_sz, [_c_i, _c_i]),
    "dmpc_pnqp_workspace_bytes": (_c_sz, [_c_i, _c_i, _c_i, _c_i]),
    "dmpc_pnqp": (_c_i, [_c_i, _c_i] + [_c_f] * 5 + [_c_i, _c_i] + [_c_f] * 5 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_mpc_step_workspace_bytes": (_c_sz, [_c_i] * 4),
    "dmpc_mpc_step_forward": (_c_i, [_c_i] * 4 + [_c_f] * 12 + [_c_i, ctypes.c_float, _c_i, _c_i, _c_i]
                              + [_c_f] * 11 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_mpc_backward_rec_workspace_bytes": (_c_sz, [_c_i] * 6),
    "dmpc_mpc_backward_rec": (_c_i, [_c_i] * 4 + [_c_f] * 7 + [_c_i, _c_i] + [_c_f] * 3 + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_mpc_forward_rec": (_c_i, [_c_i] * 4 + [_c_f] * 10 + [ctypes.c_float, _c_i] + [_c_f] * 10),
    "dmpc_mpc_forward_rec_pendulum": (_c_i, [_c_i] * 2 + [_c_f] * 8 + [ctypes.c_float] * 6 + [_c_i] + [_c_f] * 10),
    "dmpc_pendulum_rollout_linearize": (_c_i, [_c_i] * 2 + [_c_f] * 2 + [ctypes.c_float] * 5 + [_c_i] + [_c_f] * 4),
    "dmpc_mpc_step_backward": (_c_i, [_c_i] * 4 + [_c_f] * 9 + [_c_f] * 7 + [_c_f, _c_f, ctypes.c_float]
                               + [_c_f, _c_sz, _c_f, _c_f]),
    "dmpc_lin_rollout": (_c_i, [_c_i] * 4 + [_c_f] * 6),
    "dmpc_mpc_step_status": (_c_i, [_c_i] + [_c_f] * 5),
    "dmpc_box_ddp_workspace_bytes": (_c_sz, [_c_i] * 4),
    "dmpc_box_ddp": (_c_i, [_c_i] * 4 + [_c_f] * 5 + [_c_i, _c_f] + [_c_f] * 3 +
                     [ctypes.c_float, _c_i, ctypes.c_float, _c_i, ctypes.c_float, _c_i, _c_i, _c_i, _c_i] +
                     [_c_f] * 7 + [_c_sz, _c_f, _c_f]),
}

_lib = None
_lock = threading.Lock()


class DmpcError(RuntimeError):
    pass


def load(path=None):
    """dlopen the HIP library (no GPU needed for this step) and attach signatures."""
    global _lib
    with _lock:
        if _lib is not None and path is None:
            return _lib
        p = path or LIB_PATH
        if not os.path.exists(p):
            raise DmpcError(
                "%s not found at %s - build it with `python %s` (hipcc, gfx950). "
                "There is no CPU fallback." % (LIB_NAME, p, os.path.join(_HERE, "csrc", "build.py")))
        lib = ctypes.CDLL(p)
        partial = os.environ.get("DMPC_LIB_PARTIAL") == "1" and (path is not None or "DMPC_LIB" in os.environ)
        for name, (res, args) in SIGNATURES.items():
            if partial and not hasattr(lib, name):   # experiment builds of ONE translation unit (scripts/knob_variants.sh)
                continue
            fn = getattr(lib, name)      # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _check_identity(lib, p, explicit=path is not None or "DMPC_LIB" in os.environ)
        if path is None:
            _lib = lib
        return lib


def _check_identity(lib, p, explicit):
    """A library with another ABI number would receive shifted pointer arguments: refuse it.  The in-tree library must
    also be the one built from the sources in the tree (`dmpc_source_hash()` against csrc/build.py's hash of them); a
    library named explicitly (`load(path)`, `DMPC_LIB`: the variant builds of scripts/*_variants.sh) is exempt from the
    hash comparison, never from the version check.  `DMPC_SKIP_HASH_CHECK=1` drops the hash comparison for the default
    library too (a checkout whose sources are being edited while an older build is deliberately kept)."""
    ver = lib.dmpc_version()
    if ver != ABI_VERSION:
        raise DmpcError("%s has C-ABI version %d, this binding was written for %d - rebuild it (python %s)" % (
            p, ver, ABI_VERSION, os.path.join(_HERE, "csrc", "build.py")))
    if explicit or os.environ.get("DMPC_SKIP_HASH_CHECK") == "1":
        return
    # (1) the library against the stamp csrc/build.py wrote next to it: no source is read, nothing is executed
    got = lib.dmpc_source_hash().decode()
    stamp_path = p + ".srchash"
    stamp = None
    if os.path.exists(stamp_path):
        with open(stamp_path) as fh:
            stamp = fh.read().split("\n")
    if stamp is not None and stamp[0].strip() != got:
        raise DmpcError("%s (source hash %s) does not belong to the stamp next to it (%s: %s) - the library was replaced without "
                        "its stamp; rebuild it (python %s)" % (p, got, stamp_path, stamp[0].strip(), os.path.join(_HERE, "csrc", "build.py")))
    # (2) the stamp against the sources in the tree - only where there is a source tree (an installed package without the
    # kernels' sources has nothing to compare against).  What differs is named: the SOURCES (edited after the build: refused)
    # or only the build's flags / GEN_* knobs in this process's environment (the same sources: accepted)
    build_py = os.path.join(_HERE, "csrc", "build.py")
    import glob
    if not os.path.exists(build_py) or not glob.glob(os.path.join(_HERE, "csrc", "*.hip")):
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dmpc_build", build_py)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if mod.source_hash() == got:
        return
    if stamp is not None and len(stamp) > 1 and stamp[1].strip() == mod.content_hash():
        return       # same sources, built with other flags / knobs than this process's environment names
    raise DmpcError("%s was built from other SOURCES than the tree's (library %s; tree %s with this environment's flags, "
                    "sources alone %s against the stamp's %s) - rebuild it (python %s), or set DMPC_SKIP_HASH_CHECK=1 to use "
                    "it anyway" % (p, got, mod.source_hash(), mod.content_hash(),
                                   stamp[1].strip() if stamp is not None and len(stamp) > 1 else "?", build_py))


def last_kernel_name():
    """the kernel this thread's last library call launched last, as rocprofv3 lists it (asked of the HIP runtime)"""
    buf = ctypes.create_string_buffer(512)
    load().dmpc_last_kernel_name(buf, 512)
    return buf.value.decode()


def require_gpu():
    if not torch.cuda.is_available():
        raise DmpcError("no MI355X/HIP device visible: the differentiable-MPC kernels run on the GPU only "
                        "(there is no CPU fallback; the numpy oracle under oracle/ is test infrastructure)")


def ptr(t):
    """device pointer of a tensor (None -> NULL); a plain int - ctypes converts it for a c_void_p parameter"""
    if t is None:
        return None
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None):
    """torch's current HIP stream on `device` as the C-ABI's dmpc_stream_t"""
    if _raw_stream is not None and device is not None and device.index is not None:
        return _raw_stream(device.index)     # (the public accessor builds a Stream object: ~4 us per call)
    return torch.cuda.current_stream(device).cuda_stream


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def guard(device):
    """`with guard(dev):` - make `dev` the current device for the launches inside; free when it already is"""
    if device.index is None or torch.cuda.current_device() == device.index:
        return _NO_GUARD
    return torch.cuda.device(device)


_ERR = {E_BADARG: "bad argument (NULL pointer, non-positive size or a base pointer that is not 16-byte aligned)",
        E_UNSUPPORTED: "dimensions not covered by the HIP kernels",
        E_WORKSPACE: "workspace missing or too small"}


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise DmpcError("%s: %s (code %d)" % (what, _ERR.get(rc, "argument error"), rc))
    raise DmpcError("%s: HIP error %d" % (what, rc))


def f32c(t, device=None):
    """contiguous float32 device tensor, 16-byte aligned (clones when a view is misaligned)"""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        t = torch.as_tensor(t)
    if device is not None and t.device != device:
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.to(torch.float32)
    if not t.is_contiguous():
        t = t.contiguous()
    if t.data_ptr() % 16 != 0:
        t = t.clone()
    return t
