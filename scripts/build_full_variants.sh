#!/bin/bash
# Build whole-library experiment variants (all translation units) with GEN_* generator knobs / DMPC_EXTRA_FLAGS, each in
# its own copy of the sources; the libraries land in build_tmp/var/full_<name>.so (use with DMPC_LIB=...).
#   bash scripts/build_full_variants.sh "ntall:GEN_NT=plain,save,affine" "dmant:DMPC_EXTRA_FLAGS=-DDMPC_DMA_NT=1"
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/build_tmp/var
mkdir -p $OUT
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  ( W=$OUT/wf_$name; rm -rf $W; mkdir -p $W/pkg; cp -r $REPO/chainer_differentiable_mpc_amd/csrc $W/pkg/csrc; cp -r $REPO/include $W/include
    rm -rf $W/pkg/csrc/build $W/pkg/csrc/*_gen.hpp
    env $envs python $W/pkg/csrc/build.py --force --jobs 3 > $OUT/full_$name.log 2>&1 && cp $W/pkg/libdmpc_hip.so $OUT/full_$name.so
    tail -1 $OUT/full_$name.log; rm -rf $W ) &
done
wait
ls -la $OUT/full_*.so
