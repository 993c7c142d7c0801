cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_q -- python $GRAFT_REPO_ROOT/tools/qnet_layers.py 1024 u8 > /tmp/outp.txt 2>&1
tail -2 /tmp/outp.txt | cut -c1-200
f=$(find /tmp/pmc_q -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','')[:34]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'k_' not in k: continue
    print(k, {c: '%.3g'%(sorted(v)[len(v)//2]) for c,v in d.items()})
PY
