#!/bin/bash
# A/B on one box: the headline stream's two forms.  Default: the launcher takes the unrolled sweep when the previous
# launch was the same solve (back-to-back solves), the three-step loop otherwise; DMPC_NO_UNROLL=1: always the loop.
# Prints the headline (back-to-back launches), the HBM-streamed headline and DiffLqr's forward + backward loop.
cd $GRAFT_REPO_ROOT
for v in 0 1 0 1; do
  DMPC_NO_UNROLL=$v python bench.py --gpus 1 --steps 50 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); s=d['secondary']
print('DMPC_NO_UNROLL=$v: headline kernel %.2f us (frac %.3f), streamed %.2f us, DiffLqr fwd+bwd %.1f us, bwd %.1f us' % (d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], s['headline_hbm_streamed']['us_per_solve'], s['difflqr_fwd_bwd_cfg3']['us_fwd_bwd'], s['difflqr_fwd_bwd_cfg3']['us_bwd']))" | tee -a gpurun_out/unroll_ab2.txt
done
