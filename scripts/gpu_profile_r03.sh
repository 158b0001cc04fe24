#!/bin/bash
# Round-3 profiles (run on the GPU box via gpurun): rocprofv3 kernel stats of the bench and of every row at its BASELINE
# size, PMC traffic passes for the headline kernel, PMC instruction mix of the (32,8) kernel.  Summaries -> gpurun_out/.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
stats() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_r03_$name -o p -- "$@" > $OUT/prof_r03_$name.log 2>&1
  f=$(find $OUT/prof_r03_$name -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $OUT/r03_${name}_kernel_stats.csv && head -12 "$f"
}
pmc() { name=$1; filt=$2; shift; shift; ctrs="$1"; shift
  timeout 600 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $OUT/pmc_r03_$name -o p -- "$@" > $OUT/pmc_r03_$name.log 2>&1
  f=$(find $OUT/pmc_r03_$name -name "*counter_collection.csv" | head -1)
  echo "== $name ($ctrs)" | tee -a $OUT/r03_pmc_summary.txt
  [ -n "$f" ] && python3 - "$f" "$filt" <<'PY' | tee -a $OUT/r03_pmc_summary.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if sys.argv[2] in r.get('Kernel_Name', ''):
        agg[(r['Kernel_Name'][:70], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print("  %-70s %-28s n=%3d  mean=%.5g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
}
rm -f $OUT/r03_pmc_summary.txt
stats bench python3 $REPO/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-secondary
stats rows python3 $REPO/scripts/rows_workload.py
stats secondary python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline
pmc fetch lqr_asm "FETCH_SIZE" python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
pmc write lqr_asm "WRITE_SIZE" python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary
pmc w328_mix wave_mfma "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" python3 $REPO/bench.py --workload cfg5-shard --steps 3 --warmup 1 --no-cpu-baseline
pmc w328_mem wave_mfma "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" python3 $REPO/bench.py --workload cfg5-shard --steps 3 --warmup 1 --no-cpu-baseline
pmc w328_fetch wave_mfma "FETCH_SIZE" python3 $REPO/bench.py --workload cfg5-shard --steps 3 --warmup 1 --no-cpu-baseline
pmc w328_write wave_mfma "WRITE_SIZE" python3 $REPO/bench.py --workload cfg5-shard --steps 3 --warmup 1 --no-cpu-baseline
cd $REPO
# DiffLqr forward + backward loop, with saved gains and with the full second solve
cd /tmp
stats difflqr_saved python3 $REPO/scripts/difflqr_loop.py
stats difflqr_full python3 $REPO/scripts/difflqr_loop.py 0
cd $REPO
# HBM traffic of the DiffLqr training step's two kernels (saving solve, one-pass gradient): separate passes per counter
cd /tmp
rm -f $OUT/r03_pmc_difflqr.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_r03_difflqr_$ctr -o p -- python3 $REPO/scripts/difflqr_loop.py > $OUT/pmc_r03_difflqr_$ctr.log 2>&1
  f=$(find $OUT/pmc_r03_difflqr_$ctr -name "*counter_collection.csv" | head -1)
  echo "== $ctr (KiB; FETCH_SIZE counts 64-byte units as 32 on gfx950: x2)" | tee -a $OUT/r03_pmc_difflqr.txt
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/r03_pmc_difflqr.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'dmpc' in r.get('Kernel_Name', ''):
        agg[(r['Kernel_Name'][:110], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print("  %-110s %-12s n=%3d  mean=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
done
cd $REPO
