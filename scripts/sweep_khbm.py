"""(8,4) / (12,3) / (4,4 HIP): the LDS-DMA kernel with gain rows through the workspace against lqr_kernel (DMPC_NO_DMA=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from chainer_differentiable_mpc_amd import _lib
from chainer_differentiable_mpc_amd.lqr_recursion import solve_device
for (B,T,nx,nu) in ((4096,50,8,4),(4096,100,8,4),(4096,50,12,3),(4096,100,12,3),(4096,100,6,2),(4096,200,4,4)):
    p, d = bench.make_inputs(B, T, nx, nu, 0, torch.device("cuda"))
    x = torch.empty((T, B, nx), device="cuda"); u = torch.empty((T, B, nu), device="cuda")
    fn=lambda: solve_device(d["C"], d["c"], d["F"], d["f"], d["x_init"], None, T, nx, nu, out=(x, u))
    bench.settle(fn); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print("B=%d T=%d (%d,%d): %.1f us  %s" % (B,T,nx,nu,e0.elapsed_time(e1)/50*1e3,_lib.last_kernel_name()[:90]), flush=True)
