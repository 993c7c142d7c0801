#!/bin/bash
# kernel timeline of one learner update ALONE on the round-4 lock-step's engine (E = 1024): the last updates of the bench command are its `subfigures.learner_only` loop
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/trf
SRLX_NO_ROLES=1 timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trf -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --inner 20 --no-cpu-baseline --no-per-micro > /tmp/trf.log 2>&1
f=$(find /tmp/trf -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_sample_gather_wg' in r['Kernel_Name']]
a, b = idx[-6], idx[-5]
t0 = int(rows[a]['Start_Timestamp'])
print("kernels in window:", b - a, "span us: %.1f" % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:46]
    print("%8.1f -> %8.1f  dur %6.1f  q%-3s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, r.get('Queue_Id', '?'), name))
PY
