#!/bin/bash
# Round-2 measurement pass on the GPU box (run through gpurun from the repo root): the default bench line, rocprofv3 kernel stats of the
# same bench command, PMC passes (FETCH_SIZE and WRITE_SIZE separately: they do not fit one pass) for the fused convolution kernel,
# the noisy-net and Agent57_light lines.  Everything lands in gpurun_out/; the summaries worth keeping are copied to profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R
timeout 500 python bench.py > gpurun_out/r2_bench.json 2> gpurun_out/r2_bench.err; tail -c 600 gpurun_out/r2_bench.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profb /tmp/pmc_f /tmp/pmc_w
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -- python $R/bench.py --no-cpu-baseline --no-per-micro --no-subfigures > $R/gpurun_out/r2_bench_profiled.json 2>/dev/null
f=$(find /tmp/profb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r2_kernel_stats.csv && python $R/tools/kstats.py /tmp/profb 14
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/tools/fused_probe.py arm 1024 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/tools/fused_probe.py arm 1024 > /dev/null 2>&1
python - $R <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
out = {}
def mean_counter(d, counter, kernel):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                vals.append(float(r["Counter_Value"]))
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
for k, sym in (("k_convnet_fused", "k_convnet_fused<true>"), ("k_gemm", "k_gemm")):
    fe, nf = mean_counter("/tmp/pmc_f", "FETCH_SIZE", sym)
    wr, nw = mean_counter("/tmp/pmc_w", "WRITE_SIZE", sym)
    if fe is None or wr is None:
        continue
    # counters are in KiB; FETCH_SIZE reports half of a wide coalesced stream's bytes on gfx950 (MI355X_MICROARCH.md, HBM): doubled
    out[k] = {"fetch_size_kib_raw": fe, "write_size_kib_raw": wr, "launches_averaged": [nf, nw], "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
              "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs of tools/fused_probe.py arm 1024: 1024 samples per launch); "
                     "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"}
json.dump(out, open(R + "/gpurun_out/r2_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
cd $R
timeout 400 python bench.py --noisy --no-cpu-baseline --no-per-micro > gpurun_out/r2_bench_noisy.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r2_bench_noisy.json').read().strip().splitlines()[-1]);print('noisy', round(d['value']), d['ms_per_lock_step'], d['learner_updates_per_s'])"
timeout 300 python bench.py --algo ppo --steps 20 > gpurun_out/r2_bench_ppo.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r2_bench_ppo.json').read().strip().splitlines()[-1]);print('ppo', round(d['value']), d['ms_per_step'], d['learner_updates_per_s'])"
timeout 500 python bench.py --algo agent57_light --envs 1024 --capacity 200000 --steps 4 --inner 16 --warmup 1 > gpurun_out/r2_bench_agent57_light.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r2_bench_agent57_light.json').read().strip().splitlines()[-1]);print('agent57_light', round(d['value']), d['ms_per_lock_step'], d['learner_updates_per_s'])"
# kernel timelines: one lock-step of the bench loop (actors + learner), one update of the learner alone
bash $R/tools/_trace_loop.sh > $R/gpurun_out/r2_loop_timeline.txt 2>&1; head -3 $R/gpurun_out/r2_loop_timeline.txt
bash $R/tools/_trace_learner.sh > $R/gpurun_out/r2_learner_timeline.txt 2>&1; head -3 $R/gpurun_out/r2_learner_timeline.txt
timeout 900 python $R/tools/graph_replay_check.py 2>&1 | grep -v "amdgpu\|Warning\|detach\|benchmark_limit" > $R/gpurun_out/r2_graph_replay_check.txt; cat $R/gpurun_out/r2_graph_replay_check.txt
