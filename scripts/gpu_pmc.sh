#!/bin/bash
# PMC passes for the headline kernel (separate runs, --pmc only with --kernel-trace; see MI355X_MICROARCH.md).
# Usage: bash scripts/gpu_pmc.sh [tag]
TAG=${1:-r01}
REPO=$(pwd)
OUT=$REPO/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
pass() {
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_${TAG}_$name -o p -- python $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc_${TAG}_$name.log 2>&1
  f=$(find $OUT/pmc_${TAG}_$name -name "*counter_collection.csv" | head -1)
  echo "== $name: $f"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'lqr_' in r.get('Kernel_Name', ''):
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print("  %-28s n=%3d  mean=%.4g" % (k, len(v), sum(v) / len(v)))
PY
}
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
pass sq3 SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
cd $REPO
