"""CPU: bench.py and __graft_entry__.py cannot run here (no GPU), but what CAN be checked without one is: they compile,
every name they use is defined somewhere (a deleted helper shows up here, not on the GPU box), and bench.py's command
line is the one the driver uses."""
import ast
import builtins
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bound_names(node):
    """every name bound anywhere inside `node` (arguments, assignments, loops, withs, imports, nested definitions)"""
    out = set()
    for n in ast.walk(node):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            out.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for arg in a.posonlyargs + a.args + a.kwonlyargs + ([a.vararg] if a.vararg else []) + ([a.kwarg] if a.kwarg else []):
                out.add(arg.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            out.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                out.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            out.add(n.name)
    return out


@pytest.mark.parametrize("script", ["bench.py", "__graft_entry__.py"])
def test_every_name_is_defined(script):
    src = open(os.path.join(ROOT, script)).read()
    tree = ast.parse(src)
    module_names = set(dir(builtins)) | {"__file__", "__name__"}
    for n in tree.body:
        module_names |= _bound_names(n) if not isinstance(n, (ast.FunctionDef, ast.ClassDef)) else {n.name}
    for fn in [n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef))]:
        known = module_names | _bound_names(fn)
        # names of enclosing functions' locals (closures): be lenient - everything bound anywhere in the module's functions
        for outer in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            if any(inner is fn for inner in ast.walk(outer)):
                known |= _bound_names(outer)
        for n in ast.walk(fn):
            if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load):
                assert n.id in known, "%s: `%s` (line %d, in %s) is not defined" % (script, n.id, n.lineno, fn.name)


def test_bench_command_line_is_the_drivers():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--gather", "--allow-secondary-failure"):
        assert flag in r.stdout


def test_the_rccl_argument_path_up_to_init_process_group(monkeypatch):
    """bench.py's N > 1 start-up with the default backend, as far as a box without a GPU can take it: backend "nccl" (= RCCL
    on ROCm), this rank's device bound at init (device_id), the timing reductions on that device, MASTER_ADDR defaulted to
    127.0.0.1 - with torch.distributed.init_process_group and torch.cuda.set_device replaced by recorders."""
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    import bench
    calls = {}
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: calls.setdefault("set_device", d))
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **k: calls.setdefault("init", (a, k)))
    env = {}
    d, device, backend, red_dev = bench.init_distributed(world=8, rank=5, local_rank=5, environ=env)
    assert d is dist and backend == "nccl" and device == torch.device("cuda", 5) and red_dev == device
    assert calls["set_device"] == device
    a, k = calls["init"]
    assert a == ("nccl",) and k == dict(rank=5, world_size=8, device_id=device)
    assert env["MASTER_ADDR"] == "127.0.0.1"
    # the rehearsal knobs: gloo through the host, every rank on one device
    calls.clear()
    env = {"DMPC_BENCH_BACKEND": "gloo", "DMPC_BENCH_DEVICE": "0"}
    d, device, backend, red_dev = bench.init_distributed(world=2, rank=1, local_rank=1, environ=env)
    assert backend == "gloo" and device == torch.device("cuda", 0) and red_dev == torch.device("cpu")
    assert calls["init"] == (("gloo",), dict(rank=1, world_size=2))
    # one rank: no process group at all
    calls.clear()
    d, device, backend, red_dev = bench.init_distributed(world=1, rank=0, local_rank=0, environ={})
    assert d is None and "init" not in calls


def test_config5_is_a_strong_scaling_workload():
    """`--workload cfg5-shard --gpus N`: 65,536 trajectories in all whatever N is (BASELINE.json configs[4]); N = 8 gives the
    8,192-trajectory shard of the name"""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.STRONG == {"cfg5-shard": 65536}
    assert bench.STRONG["cfg5-shard"] // 8 == bench.WORKLOADS["cfg5-shard"][0] == 8192
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert '"scaling": "weak" if strong_total is None else "strong"' in src
    assert '"global_batch": world * B' in src
