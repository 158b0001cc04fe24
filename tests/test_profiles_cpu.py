"""CPU: the committed evidence is self-consistent: every kernel named in profiles/r05/bench_line.json is
a kernel the rocprofv3 summaries beside it list (so a line and its CSVs come from the same tree), and the counter traffic
bench.py quotes is the one profiles/r05/pmc_summary.txt holds."""
import csv
import glob
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R05 = os.path.join(ROOT, "profiles", "r05")


def kernel_strings(obj):
    if isinstance(obj, dict):
        for k, v in obj.items():
            if k == "kernel" and isinstance(v, str):
                yield v
            else:
                yield from kernel_strings(v)
    elif isinstance(obj, list):
        for v in obj:
            yield from kernel_strings(v)


def norm(name):
    """`void dmpc::k<8, 2, true>(dmpc::LqrArgs)` and `dmpc::k<8, 2, true>` -> `dmpc::k<8,2,true>`"""
    name = re.sub(r"^void\s+", "", name.strip())
    name = re.sub(r"\(.*\)\s*(\[clone.*\])?$", "", name)
    return name.replace(" ", "")


@pytest.mark.skipif(not os.path.exists(os.path.join(R05, "bench_line.json")), reason="profiles/r05 not generated yet")
def test_every_kernel_of_the_bench_line_is_in_the_profiler_summaries():
    line = json.load(open(os.path.join(R05, "bench_line.json")))
    named = sorted(set(norm(k) for k in kernel_strings(line)))
    assert named, "bench_line.json names no kernel"
    listed = set()
    for path in glob.glob(os.path.join(R05, "*kernel_stats.csv")):
        for row in csv.DictReader(open(path)):
            listed.add(norm(row.get("Name", "")))
    assert listed, "no rocprofv3 kernel_stats.csv next to bench_line.json"
    missing = [k for k in named if k not in listed]
    assert not missing, "kernels of bench_line.json that no profiles/r05/*kernel_stats.csv lists: %s" % missing


@pytest.mark.skipif(not os.path.exists(os.path.join(R05, "pmc_summary.txt")), reason="profiles/r05 not generated yet")
def test_quoted_counter_traffic_is_the_recorded_one():
    tj = json.load(open(os.path.join(ROOT, "profiles", "lqr_solve_traffic.json")))
    assert "r05" in tj["_source"]
    txt = open(os.path.join(R05, "pmc_summary.txt")).read()
    for key in ("headline", "cfg5-shard"):
        e = tj[key]
        assert e["hbm_bytes_per_launch"] == int(e["fetch_size_kib"] * 2048 + e["write_size_kib"] * 1024)
        assert ("%.6g" % e["fetch_size_kib"]) in txt and ("%.6g" % e["write_size_kib"]) in txt
