#!/bin/bash
# bench.py (no CPU legs) once per whole-library variant, ROUNDS times round-robin; one summary line per run.
#   bash scripts/bench_variants.sh build_tmp/var/full_a.so build_tmp/var/full_b.so      [ROUNDS=2]
for r in $(seq 1 ${ROUNDS:-2}); do
  for lib in "$@"; do
    DMPC_LIB=$(pwd)/$lib python bench.py --gpus 1 --steps 100 --warmup 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); s=d['secondary']
g=lambda k,f: s.get(k,{}).get(f,float('nan'))
print('%-32s headline %.2f us (%.3f)  one set %.2f us (%.3f)  DiffLqr f+b %.1f us, bwd %.1f us (full: %.1f / %.1f)  MPC step %.1f us  cfg2 %.3f ms  cfg4 %.3f ms  cfg5 %.3f ms' % (
  '$lib'.split('/')[-1], d['roofline']['kernel_ms']*1e3, d['roofline']['frac'], d['roofline']['kernel_ms_same_inputs']*1e3, d['roofline']['frac_same_inputs'],
  g('difflqr_fwd_bwd_cfg3','us_fwd_bwd'), g('difflqr_fwd_bwd_cfg3','us_bwd'), g('difflqr_fwd_bwd_cfg3','us_fwd_bwd_full_second_solve'), g('difflqr_fwd_bwd_cfg3','us_bwd_full_second_solve'),
  g('mpc_step_forward_cfg3','us'), g('config2_box_ddp_b128','ms_per_solve'), g('config4_box_ddp_b1024','ms_per_solve'), g('cfg5_shard','ms_per_solve')))
if 'error' in s: print('   secondary error:', s['error'])
"
  done
done
