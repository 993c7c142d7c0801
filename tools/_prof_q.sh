cd /tmp && export TMPDIR=/tmp
for B in 1024 96; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profq_$B -- python $GRAFT_REPO_ROOT/tools/qnet_layers.py $B u8 > /tmp/outq_$B.txt 2>&1
  f=$(find /tmp/profq_$B -name "*kernel_stats.csv" | head -1)
  python - "$f" $B <<'PY'
import csv,sys
print("B =", sys.argv[2])
for r in list(csv.DictReader(open(sys.argv[1])))[:7]:
    if 'k_' in r['Name']: print('  ', r['Name'].replace('(anonymous namespace)::','')[:50].ljust(50), r['Calls'].rjust(4), 'avg %7.1f min %7.1f us' % (float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
PY
done
