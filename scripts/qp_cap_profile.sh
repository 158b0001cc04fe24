cd /tmp && export TMPDIR=/tmp
for cap in 1 2 3 20; do
rm -rf /tmp/qp; timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/qp -- python3 $GRAFT_REPO_ROOT/scripts/mpc_qp_cost.py $cap > /tmp/qp.log 2>&1
f=$(find /tmp/qp -name "*kernel_stats.csv" | head -1)
echo "cap $cap: $(grep -v rocprof /tmp/qp.log | tail -1)"; grep "mpc_" $f | cut -d, -f1-4 
done
