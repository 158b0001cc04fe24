"""GEN_TIMING build of the (8,2) stream (scripts/build_variants.sh "timing!:GEN_TIMING=1"): per-wavefront s_memtime stamps
(100 MHz ticks) of prologue / backward sweep / rollout, with ONE input set re-solved (Infinity-Cache resident) and with NSET
sets in rotation (every byte from HBM) - where the HBM-streamed solve loses its time, and to which wavefronts.
    python scripts/headline_phase_spread.py build_tmp/var/lib_timing!.so"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
lib = ctypes.CDLL(os.path.abspath(sys.argv[1]))
fn = lib.dmpc_lqr_solve
vp = ctypes.c_void_p
fn.restype = ctypes.c_int
fn.argtypes = [ctypes.c_int] * 4 + [vp] * 10 + [vp, ctypes.c_size_t, vp, vp]
dev = torch.device("cuda")
B, T, nx, nu = 4096, 50, 8, 2
NSET = int(os.environ.get("NSET", "4"))
sets = [bench.make_inputs(B, T, nx, nu, s, dev)[1] for s in range(NSET)]
x = torch.empty((T, B, nx), device=dev); u = torch.empty((T, B, nu), device=dev)
info = torch.zeros(B, dtype=torch.int32, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run(d):
    rc = fn(T, B, nx, nu, d["C"].data_ptr(), d["c"].data_ptr(), d["F"].data_ptr(), d["f"].data_ptr(), d["x_init"].data_ptr(),
            None, None, None, x.data_ptr(), u.data_ptr(), None, 0, info.data_ptr(), st)
    assert rc == 0, rc


for mode in ("one set", "%d sets in rotation" % NSET):
    acc, wall = [], []
    for it in range(24):
        d = sets[0] if mode == "one set" else sets[it % NSET]
        info.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(d)
        e1.record()
        torch.cuda.synchronize()
        if it >= 8:
            acc.append(info.cpu().numpy().astype(np.int64).reshape(-1, 4)[: B // 4])
            wall.append(e0.elapsed_time(e1) * 1e3)
    a = np.stack(acc)            # [runs, waves, 4]: set-up, prologue end, backward end, end (ticks of 10 ns since the stream's start)
    pro, bwd, fwd = a[..., 1], a[..., 2] - a[..., 1], a[..., 3] - a[..., 2]
    end_max = a[..., 3].max(axis=1).mean()
    print("%-20s s_memtime counts: prologue %6.0f (max %6.0f)   backward mean %6.0f  p99 %6.0f  max %6.0f   rollout mean %5.0f max %5.0f   "
          "stream end mean %6.0f max %6.0f;  launch (events, one at a time) %.1f us -> %.2f counts per ns" % (
              mode, pro.mean(), pro.max(axis=1).mean(), bwd.mean(), np.percentile(bwd, 99), bwd.max(axis=1).mean(), fwd.mean(),
              fwd.max(axis=1).mean(), a[..., 3].mean(), end_max, np.median(wall), end_max / np.median(wall) / 1e3))

# the same, back to back as the benchmark runs it: 100 launches without a synchronisation; the stamps of the last one
for mode in ("one set", "%d sets in rotation" % NSET):
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for it in range(100):
            run(sets[0] if mode == "one set" else sets[it % NSET])
        e1.record()
        torch.cuda.synchronize()
        a = info.cpu().numpy().astype(np.int64).reshape(-1, 4)[: B // 4]     # (or-ed over the launches: an upper envelope)
    per = e0.elapsed_time(e1) * 10.0
    info.zero_()
    run(sets[0] if mode == "one set" else sets[1])
    torch.cuda.synchronize()
    a = info.cpu().numpy().astype(np.int64).reshape(-1, 4)[: B // 4]
    print("%-20s back to back: %.2f us per launch; a launch right after the loop: backward mean %6.0f max %6.0f counts, stream end max %6.0f" % (
        mode, per, (a[:, 2] - a[:, 1]).mean(), (a[:, 2] - a[:, 1]).max(), a[:, 3].max()))
