# PMC counters of the FC1 kernels (tools/fc1_probe.py): wave cycles / stalls / MFMA busy / LDS conflicts / L2 hit rate / fabric read requests
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmc_fc1
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_fc1 -- python $R/tools/fc1_probe.py > /tmp/outp.txt 2>&1
tail -4 /tmp/outp.txt | cut -c1-200
f=$(find /tmp/pmc_fc1 -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name'].replace('(anonymous namespace)::','').replace('void ','')[:40]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,d in agg.items():
    if 'k_fc1' not in k and 'k_gemm_s16' not in k and 'k_split' not in k: continue
    print(k, {c: '%.4g'%(sorted(v)[len(v)//2]) for c,v in d.items()})
PY
