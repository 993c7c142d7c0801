#!/bin/bash
# kernel timeline of one period of the LEARNER-ONLY rank of the multi-GPU job (bench.py --roles-only: ingest of 7 slabs + one update per period), under rocprofv3 --kernel-trace
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trl
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trl -- python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --roles-only > /tmp/trl.log 2>&1
f=$(find /tmp/trl -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the learner role's periods: one k_sample_gather_wg each; take a window in the middle of the run that holds slab commits (k_unpack / commit kernels)
idx = [i for i, r in enumerate(rows) if 'k_sample_gather_wg' in r['Kernel_Name']]
best = None
for j in range(len(idx) - 1):
    a, b = idx[j], idx[j + 1]
    names = [r['Kernel_Name'] for r in rows[a:b]]
    if any('commit' in n or 'unpack' in n for n in names) and not any('k_convnet_fused<true' in n for n in names):
        best = (a, b)
a, b = best
t0 = int(rows[a]['Start_Timestamp'])
print("kernels in window:", b - a, "span us: %.1f" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:50]
    print("%8.1f -> %8.1f  dur %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
PY
