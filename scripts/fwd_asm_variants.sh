#!/bin/bash
# Timing knobs of the forward line-search stream (run on the GPU box).  The variants are whole libraries built beforehand,
# out of tree, with scripts/build_full_variants.sh (e.g. "nostore:GEN_FWD_NO_STORE=1" "nocost:GEN_FWD_NO_COST=1"
# "noadv:GEN_FWD_NO_ADV=1" "nomask:GEN_FWD_NO_TRAJ_MASK=1"); the tree's own library is the first line.
#   bash scripts/fwd_asm_variants.sh build_tmp/var/full_nomask.so ...
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for lib in TREE "$@"; do
  if [ $lib = TREE ]; then unset DMPC_LIB; else export DMPC_LIB=$GRAFT_REPO_ROOT/$lib; fi
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fv && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fv -- python3 $GRAFT_REPO_ROOT/scripts/mpc_step_only.py > /tmp/fv.log 2>&1 )
  f=$(find /tmp/fv -name "*kernel_stats.csv" | head -1)
  echo "$lib: $(grep mpc_forward_asm $f | cut -d, -f4 | head -1) ns; $(tail -1 /tmp/fv.log)" | tee -a gpurun_out/fwd_variants.txt
done
