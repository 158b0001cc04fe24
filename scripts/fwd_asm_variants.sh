#!/bin/bash
# timing knobs of the forward line-search stream (run on the GPU box): each variant rebuilds mpc_api.o with a knob set
cd $GRAFT_REPO_ROOT
for v in NONE NO_STORE NO_COST NO_ADV; do   # (NO_DMA, NO_LDS leave the registers uninitialised: not run routinely)
  rm -f chainer_differentiable_mpc_amd/csrc/build/mpc_fwd_asm_gen.hpp.genhash
  if [ $v = NONE ]; then python chainer_differentiable_mpc_amd/csrc/build.py > /dev/null 2>&1; else env GEN_FWD_$v=1 python chainer_differentiable_mpc_amd/csrc/build.py > /dev/null 2>&1; fi
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/fv && timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fv -- python3 $GRAFT_REPO_ROOT/scripts/mpc_step_only.py > /tmp/fv.log 2>&1 )
  f=$(find /tmp/fv -name "*kernel_stats.csv" | head -1)
  echo "$v: $(grep mpc_forward_asm $f | cut -d, -f4 | head -1) ns; $(tail -1 /tmp/fv.log)" | tee -a gpurun_out/fwd_variants.txt
done
rm -f chainer_differentiable_mpc_amd/csrc/build/mpc_fwd_asm_gen.hpp.genhash
