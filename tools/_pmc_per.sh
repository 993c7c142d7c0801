#!/bin/bash
# PMC counters of the bulk PER walk (own pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
rm -rf /tmp/pmc_per
timeout 300 rocprofv3 --pmc $set --output-format csv -d /tmp/pmc_per -- python $GRAFT_REPO_ROOT/tools/per_probe.py quick > /tmp/outp.txt 2>&1
tail -1 /tmp/outp.txt | cut -c1-200
f=$(find /tmp/pmc_per -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:34]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'k_descend' not in k and 'k_normalise' not in k: continue
    print(k, {c: '%.4g'%(sorted(v)[len(v)//2]) for c,v in d.items()})
PY
done
