"""BASELINE.json configs[3] as the training update it names (env_dx/il_env.py:104-158, il_exp.py:213-302): learnable cost
q = sigmoid(logit), p = sqrt(q) * learn_p, box-DDP with the true pendulum WITH the gradient node, imitation loss on an
expert's controls, backward to d logit / d learn_p, RMSprop(lr=1e-2, alpha=0.5).  B=1024, T=20, 10 iLQR iterations.
Prints ONE JSON line (bench.py runs this in a process of its own and merges it into `secondary`)."""
import json, os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from chainer_differentiable_mpc_amd import IL_Env, Pendulum_Net_cost_logit, PendulumDx

device = torch.device("cuda")
Bp, Tp = int(os.environ.get("B", "1024")), 20
dx = PendulumDx()
out = {}
env = IL_Env("pendulum", lqr_iter=10, mpc_T=Tp, device=device, lazy_status=True)
np.random.seed(0)
xi = torch.as_tensor(IL_Env.sample_xinit(Bp), dtype=torch.float32, device=device)
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    with torch.no_grad():
        q_true, p_true = dx.get_true_obj()
        _, u_exp = env.mpc(env.true_dx, xi, q_true, p_true)
    net = Pendulum_Net_cost_logit(4, device=device)
    with torch.no_grad():
        net.learn_p.copy_(torch.tensor([0.05, -0.02, 0.01, 0.03], device=device))
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-2, alpha=0.5, capturable=True)      # il_exp.py:213-302

    def train_step():
        opt.zero_grad(set_to_none=True)
        _, u_pred = net(xi, env)
        loss = ((u_pred - u_exp) ** 2).mean()
        loss.backward()
        opt.step()
        return loss

    # warm-up on a side stream first (what a later graph capture asks for: allocations, library load, optimiser state)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            train_step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    replay = None
    try:                           # (c) the update captured once in a hipGraph and replayed - first: a capture_end of this
    # update after the phase-by-phase runs below (their last autograd graph still alive) has crashed inside the HIP runtime
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=True)
        with torch.cuda.graph(graph):
            static_loss = train_step()
        rp = []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                graph.replay()
            torch.cuda.synchronize()
            rp.append((time.perf_counter() - t0) / 20)
        replay = float(np.median(rp)) * 1e3
    except Exception as e:  # pragma: no cover
        replay = "capture failed: %r" % (e,)
    fw, bw = [], []
    for it in range(3 + 15):       # (a) phase by phase, each ended by a device synchronisation
        opt.zero_grad(set_to_none=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, u_pred = net(xi, env)
        loss = ((u_pred - u_exp) ** 2).mean()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss.backward()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        opt.step()
        if it >= 3:
            fw.append(t1 - t0)
            bw.append(t2 - t1)
    g_ok = bool(torch.isfinite(net.learn_q_logit.grad).all()) and float(net.learn_q_logit.grad.abs().max()) > 0
    pipe = []
    for rep in range(5):           # (b) as a training loop runs it: 20 updates, one synchronisation
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            train_step()
        torch.cuda.synchronize()
        pipe.append((time.perf_counter() - t0) / 20)
out["config4_imitation_step"] = {
    "what": "one imitation-learning update at config 4 (B=1024, T=20, 10 iLQR iterations, env_dx/il_env.py:104-158, "
            "il_exp.py:213-302): forward with the gradient node, loss on the expert's controls, backward to d logit / "
            "d learn_p (+ the RMSprop update in the last two figures).  ms_forward / ms_backward: median of 15, each "
            "phase ended by a device synchronisation; ms_update_pipelined: 20 updates per synchronisation, as a "
            "training loop runs them; ms_update_graph_replay: the update captured once in a hipGraph",
    "ms_forward": float(np.median(fw)) * 1e3, "ms_backward": float(np.median(bw)) * 1e3,
    "ms_step": float(np.median(np.add(fw, bw))) * 1e3, "ms_update_pipelined": float(np.median(pipe)) * 1e3,
    "ms_update_graph_replay": replay, "gradient_finite_and_nonzero": g_ok}
print(json.dumps(out["config4_imitation_step"]))
