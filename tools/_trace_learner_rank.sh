#!/bin/bash
# kernel timeline of ONE period of a learner-only rank (bench.py:role_timings through tools/role_probe.py: update graph with the 7168-env slab's ring commit + tree add
# on its side branch and the next update's draw behind the write-back), alone on the GPU: the window between two packed ring commits near the end of the run
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trr -- python $GRAFT_REPO_ROOT/tools/role_probe.py > /tmp/trr.log 2>&1
f=$(find /tmp/trr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_commit_step' in r['Kernel_Name']]
a, b = idx[-4], idx[-3]
# start the window at the first kernel of the period: the forward convolutions come first when the batch was pre-drawn
t0 = int(rows[a]['Start_Timestamp'])
print("kernels in window:", b - a, "span us: %.1f" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
for r in rows[a - 3:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:46]
    print("%8.1f -> %8.1f  dur %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
PY
