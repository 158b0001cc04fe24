#!/bin/bash
# Shader clock and power while one library variant solves the (32,8) shard in a loop (run on the GPU box):
#   bash scripts/clock_under_load.sh name=lib.so ...
cd $GRAFT_REPO_ROOT
for spec in "$@"; do
  NX=32 NU=8 SIZES=8192 NSET=1 REPS=400 ROUNDS=1 timeout -k 10 120 python scripts/ring_ab.py "$spec" > /tmp/load.log 2>&1 &
  pid=$!
  sleep 4
  for i in 1 2 3 4 5; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr -s ' ' | tr '\n' ';'
    echo
    sleep 0.3
  done
  wait $pid
  grep "B=" /tmp/load.log
done
