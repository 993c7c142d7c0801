#!/bin/bash
# SQ counters of the actors' convolution / first-dense-layer kernels (two passes of eight SQ counters; kernel-trace only, no other trace domain)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/pmcA /tmp/pmcB
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 4000; echo
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d /tmp/pmcA -- python $R/tools/actor_pass_probe.py 1024 30 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SALU --output-format csv -d /tmp/pmcB -- python $R/tools/actor_pass_probe.py 1024 30 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ("/tmp/pmcA", "/tmp/pmcB"):
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(d, "no counter file", glob.glob(d + "/**/*", recursive=True)[:5]); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r["Kernel_Name"]
        key = "conv" if "k_convnet_fused" in k else ("fc1" if "k_fc1_planes" in k else ("head" if "k_head" in k else None))
        if key:
            acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for key, cs in acc.items():
        print(key, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
