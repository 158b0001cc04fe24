#!/bin/bash
# Run on the GPU box (via gpurun): GPU parity tests, smoke, headline bench, rocprofv3 kernel stats.
# Usage: bash scripts/gpu_check.sh [tag]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out
[ -z "$GRAFT_REPO_ROOT" ] && OUT=$(pwd)/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu_$TAG.txt
echo "== smoke"
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5
echo "== bench"
timeout 600 python bench.py --steps 200 --warmup 20 > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.err
cat $OUT/bench_$TAG.json; tail -3 $OUT/bench_$TAG.err
echo "== rocprofv3 kernel stats"
REPO=$(pwd)
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -o lqr -- python $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
cd $REPO
find $OUT/prof_$TAG -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_$TAG -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -8 "$f"
tail -2 $OUT/prof_$TAG.log
