#!/bin/bash
# Where a pendulum box-DDP solve's time goes on the device (configs 2 and 4): rocprofv3 --kernel-trace of the device loop,
# then per solve: the sum of the kernels' own durations, the sum of the gaps between consecutive kernels of the chain, and the
# span from the first kernel's start to the last one's end.     usage (GPU box): bash scripts/ddp_chain_timeline.sh [B]
B=${1:-128}
OUT=$GRAFT_REPO_ROOT/gpurun_out/ddp_timeline_$B
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $GRAFT_REPO_ROOT/scripts/boxddp_device_loop_profile.py $B 20 > $OUT.log 2>&1
f=$(find $OUT -name "*kernel_trace.csv" | head -1)
python3 - "$f" $B <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'dmpc' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a solve's chain starts with the rollout + linearisation kernel
solves, cur = [], []
for r in rows:
    if 'pendulum_rollout_linearize' in r['Kernel_Name'] and cur:
        solves.append(cur); cur = []
    cur.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
solves.append(cur)
solves = solves[5:]          # (the first calls include lazy initialisation)
import statistics as st
ker = [sum(e - s for s, e, _ in c) / 1e3 for c in solves]
gap = [sum(c[i + 1][0] - c[i][1] for i in range(len(c) - 1)) / 1e3 for c in solves]
span = [(c[-1][1] - c[0][0]) / 1e3 for c in solves]
n = [len(c) for c in solves]
print("B=%s: %d solves; launches per solve %d; kernels' own time %.1f us, gaps between them %.1f us (%.2f us each), first start to last end %.1f us (medians)"
      % (sys.argv[2], len(solves), st.median(n), st.median(ker), st.median(gap), st.median(gap) / max(1, st.median(n) - 1), st.median(span)))
names = {}
for c in solves:
    for s, e, k in c:
        names.setdefault(k.split('(')[0][:60], []).append((e - s) / 1e3)
for k, v in sorted(names.items(), key=lambda kv: -sum(kv[1])):
    print("   %-60s x%5.1f per solve, %6.2f us each" % (k, len(v) / len(solves), sum(v) / len(v)))
PY
