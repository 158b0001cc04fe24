"""BASELINE.json configs[3]: one step of the pendulum imitation loop (env_dx/il_env.py:104-158, il_exp.py:213-302) at
batch 1024: learnable cost q = sigmoid(logit), p = sqrt(q) * learn_p, BoxDDP with the true pendulum (dynamics not
learnt: update_dynamics=False, so the gradient flows through dC, dc of MPCstep.backward), imitation loss on the
controls of an expert that knows the true cost.  Prints the wall time of forward and backward."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import BoxDDP, PendulumDx, QuadCost, TiledQuadCost
from chainer_differentiable_mpc_amd.pendulum import sample_xinit


def tile_cost(q, p, T, B):
    if os.environ.get("DMPC_NO_TILED") == "1":      # A/B: the dense cost, autograd reduces dC [T,B,4,4] to the parameters
        Q = torch.diag(q)[None, None].expand(T, B, -1, -1).contiguous()
        pv = p[None, None].expand(T, B, -1).contiguous()
        return QuadCost(Q, pv)
    return TiledQuadCost(torch.diag(q), p, T, B)    # what IL_Env.mpc builds (il_env.py:119-129)


def imitation_step(B=1024, T=20, max_iter=10, seed=0, quiet=True):
    dev = torch.device("cuda")
    dx = PendulumDx()
    q_true, p_true = (t.to(dev) for t in dx.get_true_obj())
    x0 = torch.as_tensor(sample_xinit(B, seed=seed), dtype=torch.float32, device=dev)
    kw = dict(eps=dx.mpc_eps, line_search_decay=dx.linesearch_decay, max_line_search_iter=dx.max_linesearch_iter,
              max_iter=max_iter, exit_unconverged=False, quiet=quiet)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.no_grad():   # the expert
            _, u_exp, _ = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, **kw)((x0, tile_cost(q_true, p_true, T, B), dx))
        torch.manual_seed(seed)
        logit = torch.nn.Parameter(torch.zeros(4, device=dev))            # q = sigmoid(logit)        il_env.py / pendulum_net.py:12-39
        learn_p = torch.nn.Parameter(0.1 * torch.randn(4, device=dev))    # p = sqrt(q) * learn_p
        torch.cuda.synchronize(); t0 = time.perf_counter()
        q = torch.sigmoid(logit)
        p = torch.sqrt(q) * learn_p
        solver = BoxDDP(T, dx.lower, dx.upper, B, 3, 1, None, update_dynamics=False, **kw)
        x, u, costs = solver((x0, tile_cost(q, p, T, B), dx))
        loss = ((u - u_exp) ** 2).mean()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
    return dict(loss=float(loss.detach()), fwd_ms=(t1 - t0) * 1e3, bwd_ms=(t2 - t1) * 1e3, g_logit=logit.grad.detach().cpu().numpy(),
                g_p=learn_p.grad.detach().cpu().numpy(), status=solver.status)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for rep in range(3):
        r = imitation_step(B=B)
    print("imitation step B=%d T=20: forward %.2f ms, backward %.2f ms, loss %.4f, |dlogit| %.3e, |dp| %.3e (%s)" % (
        B, r["fwd_ms"], r["bwd_ms"], r["loss"], np.abs(r["g_logit"]).max(), np.abs(r["g_p"]).max(), r["status"]))
