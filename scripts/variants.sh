#!/bin/bash
# Build single-shape (8,2) variants of the library with different -D knobs and time them (diagnostic).
# Usage (on the GPU box): bash scripts/variants.sh "name1:-DX=1 -DY=2" "name2:..."
REPO=$(pwd)
mkdir -p /tmp/var
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I$REPO/include -DDMPC_EXPERIMENT_ONLY_8_2 $flags \
      -shared -o /tmp/var/lib_$name.so $REPO/chainer_differentiable_mpc_amd/csrc/*.hip 2>&1 | grep -E "error" ) &
done
wait
for spec in "$@"; do
  name=${spec%%:*}
  echo "== $name (${spec#*:})"
  DMPC_LIB=/tmp/var/lib_$name.so python scripts/phase_timing.py 2>&1 | grep -E " us|rror"
done
