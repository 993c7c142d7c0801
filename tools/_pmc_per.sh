#!/bin/bash
# PMC counters of the bulk PER walk at 2^20 draws per call from a 1 000 000-leaf tree: FETCH_SIZE and WRITE_SIZE in two SEPARATE passes (kernel-trace only,
# no other trace domains), L2 hit / miss in a third -> gpurun_out/r5_per_pmc.json.  FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: it reports half the
# bytes of 16-byte-per-lane loads on gfx950); Infinity-Cache hits are counted by these fabric-side counters, so "traffic" is what crosses the L2, not DRAM traffic.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pp_f /tmp/pp_w /tmp/pp_h
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pp_f -- python $R/tools/per_probe.py one20 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pp_w -- python $R/tools/per_probe.py one20 > /dev/null 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pp_h -- python $R/tools/per_probe.py one20 > /dev/null 2>&1
python - $R <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
def mean(d, counter, prefix):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
            if name.startswith(prefix) and r.get("Counter_Name") == counter:
                vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 4:]
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
out = {"draws_per_call": 1 << 20, "leaves": 1_000_000, "algorithmic_bytes_per_draw": 176,
       "how": "rocprofv3 --kernel-trace --pmc <counter> -- python tools/per_probe.py one20 (three separate passes); bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"}
for key, prefix in (("walk", "k_descend_bulk"), ("finish", "k_compact_bulk")):
    fe, nf = mean("/tmp/pp_f", "FETCH_SIZE", prefix)
    wr, nw = mean("/tmp/pp_w", "WRITE_SIZE", prefix)
    hit, _ = mean("/tmp/pp_h", "TCC_HIT_sum", prefix)
    mis, _ = mean("/tmp/pp_h", "TCC_MISS_sum", prefix)
    if fe is None or wr is None:
        continue
    b = (2 * fe + wr) * 1024
    out[key] = {"kernel_prefix": prefix, "fetch_size_kib_raw": fe, "write_size_kib_raw": wr, "launches_averaged": [nf, nw], "bytes_per_launch": b,
                "bytes_per_draw": b / (1 << 20), "l2_hit_rate": (hit / (hit + mis)) if hit is not None and hit + mis > 0 else None}
if "walk" in out and "finish" in out:
    out["call_bytes_per_draw"] = out["walk"]["bytes_per_draw"] + out["finish"]["bytes_per_draw"]
    out["call_over_algorithmic"] = out["call_bytes_per_draw"] / 176
json.dump(out, open(R + "/gpurun_out/r5_per_pmc.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
