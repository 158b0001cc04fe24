"""A/B of the headline solve between two builds of libdmpc_hip.so in ONE process on one box (boxes differ by ~3 %):
    python scripts/headline_lib_ab.py build_tmp/ab/libdmpc_old.so [chainer_differentiable_mpc_amd/libdmpc_hip.so]
Only dmpc_lqr_solve is bound, so any two builds of the library can be compared."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from chainer_differentiable_mpc_amd import synthetic

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
paths = [os.path.join(root, p) for p in (sys.argv[1:] or ["build_tmp/ab/libdmpc_old.so"])]
if len(paths) == 1:
    paths.append(os.path.join(root, "chainer_differentiable_mpc_amd", "libdmpc_hip.so"))
dev = torch.device("cuda")
B, T, nx, nu = 4096, 50, 8, 2
p = synthetic.make_lqr_problem(B, T, nx, nu, seed=1)
t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)
C, c, F, f, x0 = t(p["C"]), t(p["c"]), t(p["F"]), t(p["f"]), t(p["x_init"])
x, u = torch.empty((T, B, nx), device=dev), torch.empty((T, B, nu), device=dev)
vp = ctypes.c_void_p
libs = []
for path in paths:
    lib = ctypes.CDLL(path)
    fn = lib.dmpc_lqr_solve
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] * 4 + [vp] * 10 + [vp, ctypes.c_size_t, vp, vp]
    libs.append(fn)
stream = torch.cuda.current_stream().cuda_stream

def run(fn):
    rc = fn(T, B, nx, nu, C.data_ptr(), c.data_ptr(), F.data_ptr(), f.data_ptr(), x0.data_ptr(), None, None, None,
            x.data_ptr(), u.data_ptr(), None, 0, None, stream)
    assert rc == 0, rc

def timeit(fn, reps=200):
    for _ in range(10):
        run(fn)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run(fn)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for rnd in range(4):
    print("round %d: " % rnd + "   ".join("%s %.2f us" % (os.path.basename(pth), timeit(fn)) for pth, fn in zip(paths, libs)))
