"""GPU, world_size 2: the PRODUCT path under torch.distributed.  Two fresh child processes
(spawned before anything in them touches the GPU; both on GPU 0; backend gloo, the collectives staged through the host by
dist.py) run shard_problem -> solve_device / kkt_grad_device on the HIP library -> all_gather_batch / all_reduce_param_grad
and the overlapped GatherPipeline; the gathered result must equal the UNSHARDED HIP solve bit for bit - trajectories are
independent and the kernels' arithmetic does not depend on the batch a trajectory arrives in."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("dims", [(512, 50, 8, 2), (64, 12, 32, 8), (24, 9, 5, 3)], ids=lambda d: "B%d_T%d_%dx%d" % d)
def test_two_ranks_of_the_hip_path_equal_the_unsharded_solve(tmp_path, dims):
    B, T, nx, nu = dims
    out = os.path.join(str(tmp_path), "rank0.npz")
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        try:
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out,
                                           str(B), str(T), str(nx), str(nu)], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        except OSError as e:          # a box that refuses child processes: say so, do not pretend
            for p in procs:
                p.kill()
            pytest.skip("cannot start the rank processes here: %r" % (e,))
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the rank processes did not finish within 240 s")
        logs.append(log)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    g = np.load(out)
    assert str(g["lib"]).endswith("libdmpc_hip.so") and "dmpc::" in str(g["kernel"])      # the HIP library did the solving
    assert np.array_equal(g["x_all"], g["x_full"]), "gathered x differs from the unsharded solve"
    assert np.array_equal(g["u_all"], g["u_full"])
    b0, b1 = int(g["b0"]), int(g["b1"])
    assert np.array_equal(g["dx0_local"], g["dx0_full"][b0:b1])                           # the gradient is shard-invariant too
    np.testing.assert_allclose(g["dF_sum"], g["dF_sum_full"], rtol=1e-12, atol=1e-9)     # float64 sums of identical float32 terms
    assert np.array_equal(g["piped"], g["piped_ref"]), "the overlapped gather pipeline delivered something else"


def _run_ranks(tmp_path, world, dims, backend):
    B, T, nx, nu = dims
    out = os.path.join(str(tmp_path), "rank0.npz")
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(rank), DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_gpu_worker.py"), out,
                                       str(B), str(T), str(nx), str(nu)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            log, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("the rank processes did not finish within 300 s")
        logs.append(log)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)[-3000:]
    return np.load(out)


def test_rccl_process_group_of_one_rank_drives_the_gather_pipeline(tmp_path):
    """The `nccl` (= RCCL) branch of dist.py executed for real on a one-GPU box: a process group of ONE rank on device 0, a
    fresh child process.  init_process_group("nccl", device_id=...), the device-side all_gather_into_tensor of every piece of
    GatherPipeline on its side stream (events both ways, two buffer sets in rotation over three solves), all_reduce of a
    parameter-shaped gradient, barrier, destroy - everything bench.py's N > 1 path calls except a second peer."""
    g = _run_ranks(tmp_path, 1, (256, 20, 8, 2), "nccl")
    assert str(g["backend"]) == "nccl" and "dmpc::" in str(g["kernel"])
    assert np.array_equal(g["x_all"], g["x_full"]) and np.array_equal(g["u_all"], g["u_full"])
    assert np.array_equal(g["piped"], g["piped_ref"]), "the gather pipeline over RCCL delivered something else"
    np.testing.assert_allclose(g["dF_sum"], g["dF_sum_full"], rtol=1e-12, atol=1e-9)


def test_two_ranks_over_rccl_on_two_devices(tmp_path):
    """two ranks, two devices, backend nccl (RCCL over xGMI): the gathered result equals the unsharded solve bit for bit.
    Needs two GPUs; a one-GPU box skips, saying so (torch.cuda.device_count() does not initialise the GPU in this process)."""
    import torch
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip("RCCL with two ranks needs two devices, this box has %d: the two-rank path runs over gloo on one device "
                    "(test_two_ranks_of_the_hip_path_equal_the_unsharded_solve) and RCCL with one rank "
                    "(test_rccl_process_group_of_one_rank_drives_the_gather_pipeline)" % n)
    g = _run_ranks(tmp_path, 2, (512, 50, 8, 2), "nccl")
    assert str(g["backend"]) == "nccl"
    assert np.array_equal(g["x_all"], g["x_full"]) and np.array_equal(g["u_all"], g["u_full"])
    b0, b1 = int(g["b0"]), int(g["b1"])
    assert np.array_equal(g["dx0_local"], g["dx0_full"][b0:b1])
    np.testing.assert_allclose(g["dF_sum"], g["dF_sum_full"], rtol=1e-12, atol=1e-9)
    assert np.array_equal(g["piped"], g["piped_ref"])


def test_bench_py_runs_with_two_ranks_and_the_overlapped_gather(tmp_path):
    """`bench.py --gpus 2 --gather` executed for real (everything beyond init_process_group was unexecuted code):
    two ranks on this box's one GPU through the rehearsal knobs (gloo, both ranks on device 0) - init, sharded inputs, the
    GatherPipeline in the timed region, barrier + max-over-ranks timing, rank 0's ONE JSON line with the solve / gather /
    serial / overlapped split.  The numbers are not scaling numbers (one device, host-staged collectives); the path is."""
    import json
    port = free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DMPC_BENCH_BACKEND="gloo", DMPC_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup",
                                       "1", "--workload", "pendulum", "--gather", "--gather-chunks", "2"], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            outs.append(p.communicate(timeout=400))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("bench.py with two ranks did not finish within 400 s")
    assert all(p.returncode == 0 for p in procs), "\n".join(o[1][-1500:] for o in outs)
    lines = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]      # rank 0 alone prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["global_batch"] == 2 * 1024
    assert "REHEARSAL" in j["config"]["parallelism"]
    assert j["value"] > 0 and abs(j["value"] - 2 * 1024 * 20 * 4 / (j["ms_per_step"] * 4e-3)) < 1e-6 * j["value"]
    gi = j["gather"]
    assert gi["chunks"] == 2 and gi["solve_ms"] > 0 and gi["gather_ms"] > 0 and gi["serial_ms"] >= gi["solve_ms"]
    assert gi["overlapped_ms"] == pytest.approx(j["ms_per_step"])
