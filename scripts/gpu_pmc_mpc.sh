#!/bin/bash
# Instruction mix and HBM traffic (PMC) of the two MPC-step streams at config-3 size: separate passes per counter group.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -f $OUT/r02_pmc_mpc.txt
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc_r02_mpc_$tag -o p -- python3 $REPO/scripts/mpc_step_only.py > $OUT/pmc_r02_mpc_$tag.log 2>&1
  f=$(find $OUT/pmc_r02_mpc_$tag -name "*counter_collection.csv" | head -1)
  echo "== $grp" | tee -a $OUT/r02_pmc_mpc.txt
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/r02_pmc_mpc.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'mpc_' in r.get('Kernel_Name', ''):
        agg[(r['Kernel_Name'][:72], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print("  %-72s %-24s n=%3d  mean=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
done
cd $REPO
