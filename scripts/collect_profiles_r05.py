#!/usr/bin/env python
"""Copy what scripts/gpu_profile_r05.sh left in gpurun_out/r05/ into profiles/r05/ (the tracked evidence) and refresh
profiles/lqr_solve_traffic.json (the counter traffic bench.py quotes) from its pmc_summary.txt."""
import json
import os
import re
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC, DST = os.path.join(ROOT, "gpurun_out", "r05"), os.path.join(ROOT, "profiles", "r05")
KEEP = ["parity_margins.txt", "bench_line.json", "bench_headline_kernel_stats.csv", "bench_secondary_kernel_stats.csv",
        "rows_kernel_stats.csv", "size_sweep_steady.txt", "f64_timing.txt", "pmc_summary.txt", "mpc_step_one_launch.txt", "kkt_shape_timing.txt", "mpc_shape_timing.txt", "tile16_shapes.txt", "cfg5_shard_timing.txt", "shape_floor.txt", "coupled_timing.txt"]
os.makedirs(DST, exist_ok=True)
for name in KEEP:
    p = os.path.join(SRC, name)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(DST, name))
        print("copied", name)
    else:
        print("MISSING", name)
# bench_line.json: keep the JSON line alone
bl = os.path.join(DST, "bench_line.json")
if os.path.exists(bl):
    lines = [ln for ln in open(bl).read().splitlines() if ln.startswith("{")]
    if lines:
        open(bl, "w").write(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
pm = os.path.join(DST, "pmc_summary.txt")
if os.path.exists(pm):
    txt = open(pm).read()

    def mean(section, kernel_part, counter):
        m = re.search(r"== %s .*?\n((?:  .*\n)+)" % section, txt)
        if not m:
            return None
        for ln in m.group(1).splitlines():
            if kernel_part in ln and counter in ln:
                return float(ln.split("mean=")[1])
        return None

    out = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/gpu_profile_r05.sh B); bytes = "
                   "FETCH_SIZE[KiB]*1024*2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE[KiB]*1024; "
                   "Infinity-Cache hits are included in FETCH_SIZE",
           "_source": "profiles/r05/pmc_summary.txt"}
    for key, sec, kern in (("headline", "headline", "lqr_asm_kernel<8, 2"), ("headline_f64", "f64", "lqr_f64_row_kernel<8, 2"),
                           ("cfg5-shard", "w328", "lqr_tile16_kernel<32, 8, true>")):
        f, w = mean(sec + "_fetch", kern, "FETCH_SIZE"), mean(sec + "_write", kern, "WRITE_SIZE")
        if f is not None and w is not None:
            out[key] = {"kernel": kern, "fetch_size_kib": f, "write_size_kib": w, "hbm_bytes_per_launch": int(f * 2048 + w * 1024)}
            if key == "cfg5-shard":
                out[key]["batch"] = 8192     # the counters were collected on one 8,192-trajectory shard (scripts/wave_mfma_timing.py)
    json.dump(out, open(os.path.join(ROOT, "profiles", "lqr_solve_traffic.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))
