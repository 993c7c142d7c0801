#!/bin/bash
# PMC counters of the bulk PER walk (own pass, no trace domains): L2 hit/miss and fabric-side requests
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
rm -rf /tmp/pmc_per
timeout 150 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_per -- python $GRAFT_REPO_ROOT/tools/per_probe.py quick > /tmp/outp.txt 2>&1
tail -2 /tmp/outp.txt | cut -c1-160
f=$(find /tmp/pmc_per -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'k_descend' not in k and 'k_finish_fast' not in k: continue
    # launches alternate 2^20 / 2^22 draws: report min and max
    print(k, {c: ('%.4g' % min(v), '%.4g' % max(v), len(v)) for c,v in d.items()})
PY
done
