#!/bin/bash
# per-kernel times of the Agent57_light engine (E environments; E = 16: the update dominates) from rocprofv3's kernel trace
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
E=${1:-16}
rm -rf /tmp/profa
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profa -- python $R/bench.py --algo agent57_light --envs $E --capacity ${2:-20000} --steps 6 --inner 16 --warmup 1 > /tmp/a57.json 2>/dev/null
python $R/tools/kstats.py /tmp/profa 40
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/profa/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows); calls = sum(int(r['Calls']) for r in rows)
print('kernels:', len(rows), 'calls', calls, 'total ms', tot / 1e6)
PY
