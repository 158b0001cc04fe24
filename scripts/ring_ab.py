"""A/B of builds of libdmpc_hip.so on the headline solve in ONE process on one box, Infinity-Cache resident and streamed
from HBM (input sets in rotation, > 256 MiB), at several batch sizes:
    python scripts/ring_ab.py name=path/to/lib.so [name=path ...]      [SIZES="4096 8192 32768" in the environment]
The first library is the reference: every other one must reproduce its x, u bit for bit (knob builds - names ending in
`!` - are timed only).  Only dmpc_lqr_solve is bound, so any two builds can be compared."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

dev = torch.device("cuda")
T, nx, nu = int(os.environ.get("T", "50")), int(os.environ.get("NX", "8")), int(os.environ.get("NU", "2"))
ns = nx + nu
BYTES_TS = 4 * (ns * ns + ns + nx * ns + 2 * nx + nu)      # algorithmic bytes per timestep-solve
sizes = [int(v) for v in os.environ.get("SIZES", "4096 8192 32768").split()]
vp = ctypes.c_void_p
libs = []
for spec in sys.argv[1:]:
    name, path = spec.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    fn = lib.dmpc_lqr_solve
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_int] * 4 + [vp] * 10 + [vp, ctypes.c_size_t, vp, vp]
    libs.append((name, fn))
stream = torch.cuda.current_stream().cuda_stream


WS = {}


def run(fn, d, x, u, B):
    need = T * B * nu * max(nx + nu + 1, 12) * 4   # dmpc_lqr_workspace_bytes: wide shapes / long horizons pass their gains through it
    if B not in WS:
        WS.clear()
        WS[B] = torch.empty(need, dtype=torch.uint8, device=dev)
    rc = fn(T, B, nx, nu, d["C"].data_ptr(), d["c"].data_ptr(), d["F"].data_ptr(), d["f"].data_ptr(), d["x_init"].data_ptr(),
            None, None, None, x.data_ptr(), u.data_ptr(), WS[B].data_ptr(), need, None, stream)
    assert rc == 0, rc


def timeit(fn, sets, x, u, B, reps):
    n = len(sets)
    for i in range(2 * n + 4):
        run(fn, sets[i % n], x, u, B)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        run(fn, sets[i % n], x, u, B)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for B in sizes:
    in_bytes = 4 * B * T * (ns * ns + ns + nx * ns + nx)
    nset = max(2, -(-520 * 2 ** 20 // in_bytes)) if in_bytes < 300 * 2 ** 20 else 2   # rotation larger than 2 x the cache
    nset = int(os.environ.get("NSET", "0")) or nset
    sets = [bench.make_inputs(B, T, nx, nu, 10 + s, dev)[1] for s in range(nset)]
    x, u = torch.empty((T, B, nx), device=dev), torch.empty((T, B, nu), device=dev)
    ref = None
    for name, fn in libs:
        x.zero_(); u.zero_()
        run(fn, sets[0], x, u, B)
        torch.cuda.synchronize()
        if ref is None:
            ref = (x.clone(), u.clone())
            same = "reference"
        elif name.endswith("!"):
            same = "knob build"
        else:
            same = "bit-identical" if torch.equal(x, ref[0]) and torch.equal(u, ref[1]) else "DIFFERS (max %.3g)" % float(
                max((x - ref[0]).abs().max(), (u - ref[1]).abs().max()))
        reps = int(os.environ.get("REPS", "0")) or max(10, min(200, int(3e9 / in_bytes)))
        res = []
        for rnd in range(int(os.environ.get("ROUNDS", "3"))):
            for _ in range(int(os.environ.get("CACHE_WARM", "0"))):     # make the one set Infinity-Cache resident again
                run(fn, sets[0], x, u, B)
            tc = timeit(fn, sets[:1], x, u, B, reps)
            ts = timeit(fn, sets, x, u, B, reps)
            res.append((tc, ts))
        tc = min(r[0] for r in res)
        ts = min(r[1] for r in res)
        alg = BYTES_TS * B * T
        print("B=%6d %-14s one set %7.2f us (%.3f)   %d sets in rotation %7.2f us (%.3f of 8 TB/s)   %s   [%s]"
              % (B, name, tc, alg / tc / 8e6, nset, ts, alg / ts / 8e6, same,
                 " ".join("%.1f/%.1f" % r for r in res)), flush=True)
    del sets, x, u, ref
    torch.cuda.empty_cache()
