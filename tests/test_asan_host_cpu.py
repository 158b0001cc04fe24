"""CPU, opt-in (DMPC_RUN_ASAN=1; ~6 minutes: every *_api.hip is rebuilt with its HOST pass under AddressSanitizer + UBSan):
the C-ABI's argument checking, workspace arithmetic and shape dispatch for 19 shapes x 9 horizons x 5 batch sizes, run on the
CPU through scripts/asan/asan_host_driver.cpp.  The recorded run of the round is profiles/r05/asan_host.txt, made from the sources in the tree (its hash line is checked)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(os.environ.get("DMPC_RUN_ASAN") != "1", reason="opt-in: DMPC_RUN_ASAN=1 (rebuilds the library under ASan)")
def test_host_side_of_the_c_abi_under_asan_and_ubsan(tmp_path):
    out = os.path.join(str(tmp_path), "asan.txt")
    r = subprocess.run(["bash", os.path.join(ROOT, "scripts", "asan", "run_asan_host.sh"), out], capture_output=True, text=True,
                       timeout=1800)
    txt = open(out).read()
    assert r.returncode == 0 and "0 failed" in txt and "exit code: 0" in txt, txt[-2000:]
    assert "ERROR: AddressSanitizer" not in txt and "runtime error" not in txt


def test_recorded_asan_run_is_clean_and_of_the_sources_in_the_tree():
    """the recorded run names the source set it was made from (csrc/build.py:source_hash()) - a run of another
    state of the sources is not evidence for this one"""
    import importlib.util
    import re
    p = os.path.join(ROOT, "profiles", "r05", "asan_host.txt")
    assert os.path.exists(p), "no recorded run: bash scripts/asan/run_asan_host.sh"
    txt = open(p).read()
    assert "0 failed" in txt and "exit code: 0" in txt and "ERROR: AddressSanitizer" not in txt and "runtime error" not in txt
    spec = importlib.util.spec_from_file_location("dmpc_build", os.path.join(ROOT, "chainer_differentiable_mpc_amd", "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m = re.search(r"^# library sources: ([0-9a-f]+)", txt, flags=re.M)
    assert m is not None and m.group(1) == mod.source_hash(), \
        "profiles/r05/asan_host.txt was recorded from other sources (%s, tree: %s): bash scripts/asan/run_asan_host.sh" % (
            m.group(1) if m else None, mod.source_hash())
