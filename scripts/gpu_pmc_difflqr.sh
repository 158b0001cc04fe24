#!/bin/bash
# HBM traffic (PMC) of the kernels of a DiffLqr forward + backward with saved gains: separate passes per counter.
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -f $OUT/r02_pmc_difflqr.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_r02_difflqr_$ctr -o p -- python3 $REPO/scripts/difflqr_loop.py > $OUT/pmc_r02_difflqr_$ctr.log 2>&1
  f=$(find $OUT/pmc_r02_difflqr_$ctr -name "*counter_collection.csv" | head -1)
  echo "== $ctr (KiB; FETCH_SIZE counts 64-byte units as 32 on gfx950: x2)" | tee -a $OUT/r02_pmc_difflqr.txt
  [ -n "$f" ] && python3 - "$f" <<'PY' | tee -a $OUT/r02_pmc_difflqr.txt
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'dmpc' in r.get('Kernel_Name', ''):
        agg[(r['Kernel_Name'][:96], r['Counter_Name'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print("  %-96s %-12s n=%3d  mean=%.6g" % (k[0], k[1], len(v), sum(v) / len(v)))
PY
done
cd $REPO
