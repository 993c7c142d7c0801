cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --cpu-seconds 0 > /tmp/outb.txt 2>&1
f=$(find /tmp/profb -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out; cp "$f" $GRAFT_REPO_ROOT/gpurun_out/bench_kernel_stats.csv
python - "$f" <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:32]:
    print(r['Name'].replace('(anonymous namespace)::','')[:60].ljust(60), r['Calls'].rjust(5), 'avg %7.1f min %7.1f max %7.1f  %5.1f%%' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3, float(r['Percentage'])))
PY
