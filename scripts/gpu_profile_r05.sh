#!/bin/bash
# Round-5 evidence, regenerated from the tree it runs in (on the GPU box: `gpurun -- bash scripts/gpu_profile_r05.sh A` and
# `... B`; the two halves fit gpurun's 20-minute limit separately).  Everything lands in gpurun_out/r05/, whose files are then
# copied to profiles/r05/ by scripts/collect_profiles_r05.py - nothing under profiles/r05/ is written by hand.
#   A: the GPU test suite with the parity log -> parity_margins.txt (+ the every-trajectory margins of the big sizes),
#      bench_line.json (the driver's flags), rocprofv3 kernel stats of the headline loop, of the whole bench and of every
#      row at its BASELINE size, the shape / batch / horizon sweeps, the float64 timings
#   B: PMC passes (FETCH_SIZE / WRITE_SIZE, separate runs, --pmc only with --kernel-trace) for the headline stream, the
#      float64 row kernel, the (32,8) kernel and the DiffLqr step; the (32,8) instruction mix
PART=${1:-A}
REPO=$(pwd); OUT=$REPO/gpurun_out/r05; mkdir -p $OUT; export TMPDIR=/tmp
stats() { name=$1; shift
  ( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- "$@" > $OUT/prof_$name.log 2>&1 )
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
}
pmc() { name=$1; filt=$2; shift; shift; ctrs="$1"; shift
  ( cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_$name -o p -- "$@" > $OUT/pmc_$name.log 2>&1 )
  f=$(find $OUT/pmc_$name -name "*counter_collection.csv" | head -1)
  echo "== $name ($ctrs; KiB for FETCH_SIZE / WRITE_SIZE - FETCH_SIZE counts 64-byte requests as 32 on gfx950: x2)" | tee -a $OUT/pmc_summary.txt
  [ -n "$f" ] && python3 - "$f" "$filt" <<'PY' | tee -a $OUT/pmc_summary.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r.get('Kernel_Name', ''):
        agg[(r['Kernel_Name'][:120], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print("  %-120s %-26s n=%4d  mean=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
  rm -rf $OUT/pmc_$name
}
if [ "$PART" = "A" ]; then
  rm -f $OUT/parity_log.txt
  DMPC_PARITY_LOG=$OUT/parity_log.txt timeout -k 10 900 python3 -m pytest tests -m gpu -q > $OUT/gpu_tests.log 2>&1; tail -2 $OUT/gpu_tests.log
  { python3 scripts/parity_margins.py --condense $OUT/parity_log.txt; echo; timeout -k 10 600 python3 scripts/parity_margins.py --out $OUT/parity_margins_big.txt > /dev/null 2>&1; cat $OUT/parity_margins_big.txt; } > $OUT/parity_margins.txt
  rm -f $OUT/parity_log.txt $OUT/parity_margins_big.txt
  timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err; tail -c 400 $OUT/bench_line.json
  stats bench_headline python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary
  stats bench_secondary python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline
  stats rows python3 $REPO/scripts/rows_workload.py
  { timeout -k 10 300 python3 scripts/size_sweep.py; SHAPES="4x2,3x1,6x2,4x4,8x4,12x3,6x3,13x2,9x4,11x4,5x5,12x4,12x8,16x4,16x8,13x3,10x6,15x7,20x6,17x4,24x8,24x4,32x4" timeout -k 10 300 python3 scripts/size_sweep.py; } > $OUT/size_sweep_steady.txt 2>&1
  timeout -k 10 300 python3 scripts/f64_timing.py > $OUT/f64_timing.txt 2>&1
  timeout -k 10 300 python3 scripts/mpc_step_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/mpc_step_one_launch.txt
  timeout -k 10 300 python3 scripts/kkt_shape_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/kkt_shape_timing.txt
  timeout -k 10 300 python3 scripts/mpc_shape_timing.py 2>&1 | grep -v amdgpu.ids > $OUT/mpc_shape_timing.txt
  timeout -k 10 300 python3 scripts/tile16_shapes.py 4096 2>&1 | grep -v amdgpu.ids > $OUT/tile16_shapes.txt
  timeout -k 10 300 python3 scripts/wave_mfma_timing.py 8192 2>&1 | grep -v amdgpu.ids > $OUT/cfg5_shard_timing.txt
  timeout -k 10 600 python3 scripts/coupled_timing.py 2>&1 | grep -v "Warning\|amdgpu.ids" > $OUT/coupled_timing.txt
  ls -la $OUT
else
  rm -f $OUT/pmc_summary.txt
  B="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
  pmc headline_fetch lqr_asm "FETCH_SIZE" $B
  pmc headline_write lqr_asm "WRITE_SIZE" $B
  pmc f64_fetch lqr_f64_row "FETCH_SIZE" python3 $REPO/scripts/f64_timing.py headline
  pmc f64_write lqr_f64_row "WRITE_SIZE" python3 $REPO/scripts/f64_timing.py headline
  # one 8,192-trajectory shard of config 5 (the fused solve of scripts/wave_mfma_timing.py: backward sweep, forward sweep, fused)
  W="python3 $REPO/scripts/wave_mfma_timing.py 8192"
  pmc w328_fetch "lqr_tile16_kernel<32, 8, true>" "FETCH_SIZE" $W
  pmc w328_write "lqr_tile16_kernel<32, 8, true>" "WRITE_SIZE" $W
  pmc w328_mix "lqr_tile16_kernel<32, 8, true>" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" $W
  pmc w328_mix2 "lqr_tile16_kernel<32, 8, true>" "SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" $W
  # (8,4) and (4,4) at B = 4096, T = 50: executed instructions and wavefront cycles per step (verdict r04 item 7; scripts/shape_floor.py)
  export SHAPES=8x4,4x4
  S="python3 $REPO/scripts/size_sweep.py"
  pmc s84_mix "lqr_kernel<8, 4, 16" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $S
  pmc s44_mix "lqr_asm_kernel<4, 4" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $S
  unset SHAPES
  B="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
  pmc s82_mix "lqr_asm_kernel<8, 2" "SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" $B
  python3 $REPO/scripts/shape_floor.py $OUT/pmc_summary.txt > $OUT/shape_floor.txt
  pmc difflqr_fetch dmpc "FETCH_SIZE" python3 $REPO/scripts/difflqr_loop.py
  pmc difflqr_write dmpc "WRITE_SIZE" python3 $REPO/scripts/difflqr_loop.py
  cat $OUT/pmc_summary.txt | tail -40
fi
