#!/bin/bash
# The GPU suite once per family switch: tests that assert WHICH kernel ran fail by design; anything else
# failing is a fallback that has rotted.  Prints, per switch, the pass / fail counts and the failing tests' assertion lines
# that do not mention a path / availability check.
cd $GRAFT_REPO_ROOT
for sw in DMPC_NO_WAVE_MFMA DMPC_NO_STASH DMPC_NO_ADJOINT DMPC_NO_SAVED_GAINS DMPC_NO_SPEC4 DMPC_NO_SPEC_LS DMPC_NO_CONTAINER DMPC_NO_MPC_ASM DMPC_NO_ASM DMPC_NO_DMA DMPC_NO_MPC_DMA; do
  env $sw=1 timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/sweep_$sw.log 2>&1
  echo "== $sw: $(tail -1 gpurun_out/sweep_$sw.log)"
  grep "^E  " gpurun_out/sweep_$sw.log | grep -v "solve_path\|_FuncPtr\|CDLL\|_lib.load\|assert [0-9] == [0-9]\|saving solve serves\|is not None\|unpack non-iterable\|assert False\|kernel_family\|where " | sort | uniq -c | sort -rn | head -8
done
