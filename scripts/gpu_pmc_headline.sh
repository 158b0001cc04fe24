#!/bin/bash
# PMC traffic passes for the headline kernel only (separate --pmc runs, as the guide prescribes) -> gpurun_out/r02_pmc_headline.txt
REPO=$(pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
rm -f $OUT/r02_pmc_headline.txt
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_r02h_$ctr -o p -- python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $OUT/pmc_r02h_$ctr.log 2>&1
  f=$(find $OUT/pmc_r02h_$ctr -name "*counter_collection.csv" | head -1)
  python3 - "$f" $ctr <<'PY' | tee -a $OUT/r02_pmc_headline.txt
import csv, sys
v = [float(r['Counter_Value']) for r in csv.DictReader(open(sys.argv[1])) if 'lqr_asm' in r.get('Kernel_Name', '') and r['Counter_Name'] == sys.argv[2]]
print("lqr_asm_kernel<8,2,...> (v11) %s n=%d mean=%.6g" % (sys.argv[2], len(v), sum(v) / max(len(v), 1)))
PY
done
