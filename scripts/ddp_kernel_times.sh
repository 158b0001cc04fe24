#!/bin/bash
# per-kernel average durations of the pendulum box-DDP device loop (config 2): rocprofv3 --kernel-trace --stats
# usage (on the GPU box): bash scripts/ddp_kernel_times.sh [B] [tag]
B=${1:-128}; TAG=${2:-ddp}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/scripts/boxddp_host_gpu_split.py $B > $OUT.log 2>&1
grep "wall" $OUT.log
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name']
    if 'dmpc' in n:
        print("%-70s calls %5s  avg %8.2f us  min %8.2f" % (n[:70], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3))
PY
