"""MPCstep.forward (dmpc_mpc_step_forward) at config-3 size - B=4096, T=50, (8,2), bounds +-0.5, the problem bench.py's
`mpc_step_forward_cfg3` times: HIP-event time per call of the one-launch form against the two launches
(DMPC_NO_MPC_FUSED=1), each in a child process of its own, plus a bitwise comparison of everything the two forms return."""
import hashlib, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    import torch
    import bench
    from chainer_differentiable_mpc_amd import LinDx, _lib
    from chainer_differentiable_mpc_amd.util import get_traj
    device = torch.device("cuda")
    B, T, nx, nu = int(os.environ.get("B", 4096)), int(os.environ.get("T", 50)), 8, 2
    p, d = bench.make_inputs(B, T, nx, nu, 0, device)
    torch.manual_seed(0)
    un = (0.5 * torch.randn((T, B, nu), device=device)).clamp(-0.5, 0.5)
    xn = get_traj(T, un, d["x_init"], LinDx(d["F"], d["f"]))
    lo, hi = torch.full((T, B, nu), -0.5, device=device), torch.full((T, B, nu), 0.5, device=device)
    lib = _lib.load()
    f32 = dict(dtype=torch.float32, device=device)
    Ks, ks = torch.empty((T, B, nu, nx), **f32), torch.empty((T, B, nu), **f32)
    xo, uo, u1 = torch.empty((T, B, nx), **f32), torch.empty((T, B, nu), **f32), torch.empty((T, B, nu), **f32)
    costs, old, al = torch.empty((B,), **f32), torch.empty((B,), **f32), torch.empty((B,), **f32)
    objs = torch.empty((T, B), **f32)
    nqp, nls = torch.empty((B,), dtype=torch.int32, device=device), torch.empty((B,), dtype=torch.int32, device=device)
    info = torch.zeros((B,), dtype=torch.int32, device=device)
    need = lib.dmpc_mpc_step_workspace_bytes(T, B, nx, nu)
    ws = torch.empty(need, dtype=torch.uint8, device=device)
    P = _lib.ptr

    def mpc_fwd():
        rc = lib.dmpc_mpc_step_forward(T, B, nx, nu, P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), P(un), P(xn), P(lo), P(hi),
                                       P(d["C"]), P(d["c"]), P(d["F"]), P(d["f"]), 1, 0.2, 5, 20, 0, P(xo), P(uo), P(Ks),
                                       P(ks), P(costs), P(old), P(al), P(objs), P(u1), P(nqp), P(nls), P(ws), need,
                                       P(info), _lib.stream_ptr(device))
        assert rc == 0, rc

    ts = [bench.event_time(mpc_fwd, 30) * 1e6 for _ in range(3)]
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for a in (xo, uo, Ks, ks, costs, old, al, objs, u1, nqp, nls, info):
        h.update(a.cpu().numpy().tobytes())
    print(json.dumps({"us": [round(t, 2) for t in ts], "sha256_of_outputs": h.hexdigest()[:16],
                      "qp_passes_per_timestep": float(nqp.float().mean()) / T, "ls_passes": float(nls.float().mean())}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for env in ({}, {"DMPC_NO_MPC_FUSED": "1"}, {}, {"DMPC_NO_MPC_FUSED": "1"}):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, capture_output=True, text=True, timeout=300)
            print("one launch " if not env else "two launches", r.stdout.strip()[-400:], "" if r.returncode == 0 else r.stderr.strip()[-300:], flush=True)
