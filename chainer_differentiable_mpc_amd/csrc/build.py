#!/usr/bin/env python
"""Build libdmpc_hip.so (gfx950) in-tree: one hipcc -c per translation unit, in parallel, then link.

    python chainer_differentiable_mpc_amd/csrc/build.py [--force] [--jobs N]

hipcc cross-compiles without a GPU.  The .so lands next to the Python package so that it travels
to the GPU box with the repo snapshot and shows up as loaded native code.

Staleness is decided by CONTENT, not by mtimes: every object carries the SHA-256 of what it was
compiled from (its .hip, every header, the flags), and the library carries the hash of the whole
source set twice - in `libdmpc_hip.so.srchash` next to it and inside the code (`dmpc_source_hash()`),
so a library that does not belong to the sources in the tree is detected (`is_current()`), whatever
the file times say.  Both generated headers (`lqr_asm_gen.hpp`, `dpp_blocks_gen.hpp`) are written here
from their generators and are not kept in git.
"""
import argparse
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)
OUT = os.path.join(PKG, "libdmpc_hip.so")
STAMP = OUT + ".srchash"
OBJ_DIR = os.path.join(HERE, "build")
ARCH = "gfx950"
BASE_FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize"] \
    + os.environ.get("DMPC_EXTRA_FLAGS", "").split()   # experiment knobs
FLAGS = BASE_FLAGS + ["-I" + os.path.join(ROOT, "include")]
HASHED_FLAGS = BASE_FLAGS + ["-Iinclude"]     # what the hashes see: the same in every checkout path
GENERATED = {"lqr_asm_gen.hpp": "gen_lqr_asm.py", "dpp_blocks_gen.hpp": "gen_dpp_blocks.py",
             "mpc_fwd_asm_gen.hpp": "gen_mpc_fwd_asm.py", "dpp_blocks_f64_gen.hpp": "gen_dpp_blocks_f64.py",
             "dpp_blocks_wide_gen.hpp": "gen_dpp_blocks_wide.py"}
HASH_TU = "lu_api.hip"      # the translation unit that defines dmpc_source_hash()


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found - cannot build the HIP kernels")
    return exe


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in sorted(paths, key=os.path.basename):
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def hand_written_sources():
    """everything a human edits: .hip, hand-written headers, the generators, the public header, this script"""
    srcs = glob.glob(os.path.join(HERE, "*.hip")) + glob.glob(os.path.join(ROOT, "include", "*.h")) \
        + [p for p in glob.glob(os.path.join(HERE, "*.hpp")) if os.path.basename(p) not in GENERATED] \
        + generators() + [os.path.abspath(__file__)]
    return srcs


def gen_env():
    """the GEN_* knobs of the generators (timing experiments; they change the generated streams)"""
    return repr(sorted((k, v) for k, v in os.environ.items() if k.startswith("GEN_")))


def generators():
    """every generator script: they import from each other (gen_mpc_fwd_asm.py uses gen_lqr_asm.py's emitter)"""
    return sorted(glob.glob(os.path.join(HERE, "gen_*.py")))


def source_hash():
    return _sha(hand_written_sources(), " ".join(HASHED_FLAGS) + gen_env())[:32]


def content_hash():
    """the hand-written sources alone: no flags, no GEN_* knobs - what tells "the sources were edited" from "built with knobs"."""
    return _sha(hand_written_sources(), "")[:32]


def is_current():
    """the library in the tree was built from the sources in the tree"""
    if not (os.path.exists(OUT) and os.path.exists(STAMP)):
        return False
    return open(STAMP).read().split()[0] == source_hash()


def generate(force=False):
    for out_name, gen_name in GENERATED.items():
        gen = os.path.join(HERE, gen_name)
        out = os.path.join(HERE, out_name)
        stamp = os.path.join(OBJ_DIR, out_name + ".genhash")
        want = _sha(generators(), gen_env())     # the whole generator set and its knobs, not this script alone
        if force or not os.path.exists(out) or not os.path.exists(stamp) or open(stamp).read() != want:
            r = subprocess.run([sys.executable, gen], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("%s failed:\n%s\n%s" % (gen_name, r.stdout, r.stderr))
            open(stamp, "w").write(want)


def include_closure(src):
    """the files a translation unit really includes (quoted includes, followed recursively)"""
    import re
    seen, todo = set(), [src]
    while todo:
        p = os.path.normpath(todo.pop())
        if p in seen or not os.path.exists(p):
            continue
        seen.add(p)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(p).read(), flags=re.M):
            todo.append(os.path.join(os.path.dirname(p), inc))
    return sorted(seen)


def compile_one(src, force, src_hash):
    base = os.path.basename(src)
    obj = os.path.join(OBJ_DIR, base[:-4] + ".o")
    stamp = obj + ".hash"
    deps = include_closure(src)
    flags = list(FLAGS)
    if base == HASH_TU:
        flags.append('-DDMPC_SOURCE_HASH="%s"' % src_hash)
    want = _sha(deps, " ".join(f for f in flags if not f.startswith("-I")))
    if force or not os.path.exists(obj) or not os.path.exists(stamp) or open(stamp).read() != want:
        cmd = [hipcc()] + flags + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        open(stamp, "w").write(want)
        return obj, True
    return obj, False


def build(force=False, jobs=None, verbose=True):
    os.makedirs(OBJ_DIR, exist_ok=True)
    generate(force)
    src_hash = source_hash()
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    jobs = jobs or min(len(srcs), max(1, (os.cpu_count() or 2) - 1))
    with concurrent.futures.ThreadPoolExecutor(jobs) as ex:
        res = list(ex.map(lambda s: compile_one(s, force, src_hash), srcs))
    objs = [o for o, _ in res]
    rebuilt = sum(1 for _, r in res if r)
    if force or rebuilt or not is_current():
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        # line 1: what dmpc_source_hash() returns (sources + flags + GEN_* knobs); line 2: the sources alone; line 3: the knobs
        open(STAMP, "w").write("%s\n%s\n%s\n" % (src_hash, content_hash(), " ".join(HASHED_FLAGS) + " " + gen_env()))
    if verbose:
        print("built", OUT, "(%d translation units, %d recompiled, source hash %s)" % (len(srcs), rebuilt, src_hash))
    return OUT


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=None)
    a = ap.parse_args()
    build(a.force, a.jobs)
