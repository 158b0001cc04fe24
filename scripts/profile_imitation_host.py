import cProfile, pstats, os, sys, warnings, io
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from chainer_differentiable_mpc_amd import IL_Env, Pendulum_Net_cost_logit, PendulumDx
device = torch.device("cuda")
Bp, Tp = 1024, 20
dx = PendulumDx()
env = IL_Env("pendulum", lqr_iter=10, mpc_T=Tp, device=device, lazy_status=os.environ.get("LAZY", "1") == "1")
np.random.seed(0)
xi = torch.as_tensor(IL_Env.sample_xinit(Bp), dtype=torch.float32, device=device)
warnings.simplefilter("ignore")
with torch.no_grad():
    q_true, p_true = dx.get_true_obj()
    _, u_exp = env.mpc(env.true_dx, xi, q_true, p_true)
net = Pendulum_Net_cost_logit(4, device=device)
with torch.no_grad():
    net.learn_p.copy_(torch.tensor([0.05, -0.02, 0.01, 0.03], device=device))
opt = torch.optim.RMSprop(net.parameters(), lr=1e-2, alpha=0.5, capturable=True)
def train_step():
    opt.zero_grad(set_to_none=True)
    _, u_pred = net(xi, env)
    loss = ((u_pred - u_exp) ** 2).mean()
    loss.backward()
    opt.step()
    return loss
for _ in range(10): train_step()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(100): train_step()
torch.cuda.synchronize()
print("ms per update (pipelined):", (time.perf_counter() - t0) * 10)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): train_step()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats(os.environ.get("SORT", "cumulative")).print_stats(45); print(s.getvalue()[:9000])
