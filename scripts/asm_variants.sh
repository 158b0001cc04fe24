#!/bin/bash
# Timing experiments on the generated LQR instruction stream (GPU box): regenerate lqr_asm_gen.hpp with GEN_* knobs
# (see gen_lqr_asm.py), build a single-shape library per variant and time it.  Knob builds give wrong results.
# Usage: bash scripts/asm_variants.sh "name1:GEN_X=1 GEN_Y=1" "base:" ...   [ARGS="B T nx nu" in the environment]
REPO=$(pwd)
mkdir -p /tmp/var
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( W=/tmp/var/w_$name; rm -rf $W; mkdir -p $W/pkg; cp -r $REPO/chainer_differentiable_mpc_amd/csrc $W/pkg/csrc; cp -r $REPO/include $W/include
    env $envs python $W/pkg/csrc/gen_lqr_asm.py --out $W/pkg/csrc/lqr_asm_gen.hpp > /dev/null
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I$W/include -DDMPC_EXPERIMENT_ONLY_8_2 \
      -shared -o /tmp/var/lib_$name.so $W/pkg/csrc/*.hip 2>&1 | grep -E "error" ) &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name (${spec#*:})"
  DMPC_LIB=/tmp/var/lib_$name.so python ${SCRIPT:-scripts/phase_timing.py} $ARGS 2>&1 | grep -E "${PATTERN:-fused|rror|cycles}"
done
