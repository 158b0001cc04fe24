#!/bin/bash
# Knob builds of ONE shape of the LQR solve (lqr_api.hip + lqr_wave_api.hip only, a few seconds each, in parallel) timed
# against each other on the box they were built on: which part of a kernel's time is which (timing knobs give wrong results
# on purpose).      SHAPE=4_4 ARGS="4096 50 4 4" bash scripts/knob_variants.sh "base:" "nofwd:-DDMPC_TIMING_SKIP_FWD" ...
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/knobvar; mkdir -p $OUT
CS=$REPO/chainer_differentiable_mpc_amd/csrc
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I$REPO/include -DDMPC_EXPERIMENT_ONLY_${SHAPE:-4_4} $flags \
      -shared -o $OUT/lib_$name.so $CS/lqr_api.hip $CS/lqr_wave_api.hip $CS/lu_api.hip 2>&1 | grep -E "error" ) &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name (${spec#*:})"
  DMPC_LIB=$OUT/lib_$name.so DMPC_LIB_PARTIAL=1 python $REPO/scripts/phase_timing.py ${ARGS:-4096 50 4 4} 2>&1 | grep -E " us|rror"
done
