#!/usr/bin/env python
"""Build libdmpc_hip.so (gfx950) in-tree: one hipcc -c per translation unit, in parallel, then link.

    python chainer_differentiable_mpc_amd/csrc/build.py [--force] [--jobs N]

hipcc cross-compiles without a GPU.  The .so lands next to the Python package so that it travels
to the GPU box with the repo snapshot and shows up as loaded native code.
"""
import argparse
import concurrent.futures
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libdmpc_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize",
         "-I" + os.path.join(ROOT, "include")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found - cannot build the HIP kernels")
    return exe


def newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def compile_one(src, force):
    obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
    deps = [src] + glob.glob(os.path.join(HERE, "*.hpp")) + glob.glob(os.path.join(ROOT, "include", "*.h")) \
        + [os.path.abspath(__file__)]
    if force or newer(obj, deps):
        cmd = [hipcc()] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj


def generate(force=False):
    """lqr_asm_gen.hpp (3.8 MB of generated instruction streams) is not kept in git: gen_lqr_asm.py writes it here"""
    gen = os.path.join(HERE, "gen_lqr_asm.py")
    out = os.path.join(HERE, "lqr_asm_gen.hpp")
    if force or newer(out, [gen]):
        r = subprocess.run([sys.executable, gen], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gen_lqr_asm.py failed:\n%s\n%s" % (r.stdout, r.stderr))


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    generate(force)
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(lambda s: compile_one(s, force), srcs))
    if force or newer(OUT, objs):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if verbose:
        print("built", OUT, "(%d translation units)" % len(srcs))
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    build(a.force, a.jobs)
