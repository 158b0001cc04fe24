#!/bin/bash
# Build single-shape (8,2) experiment variants of the LQR solve library (lqr_api.hip only: dmpc_lqr_solve & co.) with
# GEN_* knobs of gen_lqr_asm.py, here or on the GPU box; the libraries land in build_tmp/var/lib_<name>.so and are
# timed against each other by scripts/ring_ab.py.  Knob builds may give wrong results on purpose.
#   bash scripts/build_variants.sh "base:" "d4:GEN_RING_DEPTH=4" "d4nt:GEN_RING_DEPTH=4 GEN_NT=1"
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/build_tmp/var
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( W=$OUT/w_$name; rm -rf $W; mkdir -p $W/pkg; cp -r $REPO/chainer_differentiable_mpc_amd/csrc $W/pkg/csrc; cp -r $REPO/include $W/include
    rm -rf $W/pkg/csrc/build
    env $envs python $W/pkg/csrc/gen_lqr_asm.py --out $W/pkg/csrc/lqr_asm_gen.hpp > /dev/null || exit 1
    # SHAPE=32_8: the (32,8) wavefront-per-trajectory kernels (lqr_wave_api.hip) instead of the (8,2) streams;
    # per-variant compiler flags ride in the spec as CFLAGS=-D...
    cfl=$(echo $envs | tr ' ' '\n' | grep '^CFLAGS=' | sed 's/^CFLAGS=//')
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I$W/include -DDMPC_EXPERIMENT_ONLY_${SHAPE:-8_2} ${EXTRA_FLAGS} $cfl \
      -shared -o $OUT/lib_$name.so $W/pkg/csrc/lqr_api.hip $([ "${SHAPE:-8_2}" = 32_8 ] && echo $W/pkg/csrc/lqr_wave_api.hip) 2>&1 | grep -E "error|warning: v" ; rm -rf $W ) &
done
wait
ls -la $OUT/*.so
