#!/bin/bash
# kernel timeline of one learner update (learner alone: 16 envs), from rocprofv3's kernel trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --envs 16 --capacity 200000 --steps 2 --warmup 1 --inner 20 --no-cpu-baseline --no-per-micro --no-subfigures $EXTRA > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# find the last k_sample_wg .. next k_sample_wg window
idx = [i for i, r in enumerate(rows) if 'k_sample_gather_wg' in r['Kernel_Name'] or 'k_sample_wg' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
print("kernels in window:", b - a, "span us: %.1f" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:46]
    print("%8.1f  dur %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
PY
