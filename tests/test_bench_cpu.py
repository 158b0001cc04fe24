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
