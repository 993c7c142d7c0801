#!/bin/bash
# Round-6 measurement pass on the GPU box.  Everything lands in gpurun_out/r6_*; the summaries to be judged are copied into profiles/ by hand.
#   1. kernel-trace + stats of the default bench command            -> r6_kernel_stats.csv, r6_bench_profiled.json
#   2. FETCH_SIZE / WRITE_SIZE (two SEPARATE --pmc passes, kernel-trace only) of tools/actor_pass_probe.py:
#      the actors' policy pass exactly as the engine launches it     -> r6_pmc_traffic.json (keys: k_convnet_fused, fc1 = k_fc1_planes_h, k_head)
#   3. kernel-trace + stats of the bulk PER probe                    -> r6_per_kernel_stats.csv
#   4. the plain bench line (no profiler), after step 2 so that it reads that PMC file -> r6_bench.json
#   5. the other bench lines, replay-determinism checks, phase clocks, free-running lock-step phases, the update's timeline alone, SQ counters, PER counters, A57 kernel stats
# Kernels are matched by PREFIX (k_convnet_fused<true = the chip-filling instantiation whatever its further template arguments).
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/profb /tmp/pmc_f /tmp/pmc_w /tmp/profper
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -- python $R/bench.py --no-cpu-baseline --no-per-micro --no-subfigures > $R/gpurun_out/r6_bench_profiled.json 2>/dev/null
f=$(find /tmp/profb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $R/gpurun_out/r6_kernel_stats.csv && python $R/tools/kstats.py /tmp/profb 16
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -- python $R/tools/actor_pass_probe.py 1024 30 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -- python $R/tools/actor_pass_probe.py 1024 30 > /dev/null 2>&1
python - $R <<'PY'
import csv, glob, json, sys
R = sys.argv[1]
out = {}
def mean_counter(d, counter, prefix):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            name = r.get("Kernel_Name", "").replace("(anonymous namespace)::", "").replace("void ", "")
            if name.startswith(prefix) and r.get("Counter_Name") == counter:
                vals.append(float(r["Counter_Value"]))
    vals = vals[len(vals) // 4:]  # drop the warm-up launches
    return (sum(vals) / len(vals), len(vals)) if vals else (None, 0)
for key, prefix in (("k_convnet_fused", "k_convnet_fused<true"), ("fc1", "k_fc1_planes"), ("k_gemm_s16", "k_gemm_s16<APlain"), ("k_head", "k_head")):
    fe, nf = mean_counter("/tmp/pmc_f", "FETCH_SIZE", prefix)
    wr, nw = mean_counter("/tmp/pmc_w", "WRITE_SIZE", prefix)
    if fe is None or wr is None:
        continue
    # counters are in KiB; FETCH_SIZE reports half of a wide coalesced stream's bytes on gfx950 (MI355X_MICROARCH.md, HBM section): doubled
    out[key] = {"kernel_prefix": prefix, "fetch_size_kib_raw": fe, "write_size_kib_raw": wr, "launches_averaged": [nf, nw], "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
                "how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate runs of tools/actor_pass_probe.py 1024 30: 1024 rows per launch); "
                       "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024"}
json.dump(out, open(R + "/gpurun_out/r6_pmc_traffic.json", "w"), indent=1)
print(json.dumps({k: round(v["hbm_bytes_per_launch"] / 1e6, 2) for k, v in out.items()}), "MB per launch")
PY
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profper -- python $R/tools/per_probe.py quick > $R/gpurun_out/r6_per_probe.log 2>/dev/null
f=$(find /tmp/profper -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep -v "at::native" "$f" | head -14 > $R/gpurun_out/r6_per_kernel_stats.csv
cat $R/gpurun_out/r6_per_probe.log; python $R/tools/kstats.py /tmp/profper 8
cd $R
# the PMC file the bench line cites must be the one measured on THIS build
mkdir -p profiles; cp gpurun_out/r6_pmc_traffic.json profiles/r6_pmc_traffic.json
timeout 600 python bench.py > gpurun_out/r6_bench.json 2> gpurun_out/r6_bench.err; tail -c 1500 gpurun_out/r6_bench.json; echo
# the other bench lines of the round and the replay-determinism check
timeout 300 python bench.py --algo ppo --steps 30 > gpurun_out/r6_bench_ppo.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r6_bench_ppo.json').read().strip().splitlines()[-1]);print('ppo', round(d['value']), d['ms_per_step'], d['learner_updates_per_s'])"
timeout 500 python bench.py --algo agent57_light --envs 1024 --capacity 200000 --steps 4 --inner 16 --warmup 1 > gpurun_out/r6_bench_agent57_light.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r6_bench_agent57_light.json').read().strip().splitlines()[-1]);print('agent57_light', round(d['value']), d['ms_per_lock_step'], d['learner_updates_per_s'])"
timeout 400 python bench.py --noisy --no-cpu-baseline --no-per-micro > gpurun_out/r6_bench_noisy.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r6_bench_noisy.json').read().strip().splitlines()[-1]);print('noisy', round(d['value']), d['ms_per_lock_step'], d['learner_updates_per_s'])"
timeout 900 python tools/graph_replay_check.py 2>&1 | grep -v "amdgpu\|Warning\|detach\|benchmark_limit" > gpurun_out/r6_graph_replay_check.txt; cat gpurun_out/r6_graph_replay_check.txt
timeout 300 python tools/ppo_replay_bisect.py 2>&1 | grep -v amdgpu > gpurun_out/r6_ppo_replay_bisect.txt; cat gpurun_out/r6_ppo_replay_bisect.txt
python tools/fused_phases.py 2>&1 | tail -10 > gpurun_out/r6_fused_phases.txt; cat gpurun_out/r6_fused_phases.txt
(python tools/qnet_accuracy.py; SRLX_CONV1_F32=1 SRLX_CONV23_F32=1 SRLX_FC1_F32=1 python tools/qnet_accuracy.py) 2>&1 | grep -v amdgpu > gpurun_out/r6_qnet_accuracy.txt; cat gpurun_out/r6_qnet_accuracy.txt
bash tools/_trace_loop.sh > gpurun_out/r6_loop_timeline.txt 2>&1; head -3 gpurun_out/r6_loop_timeline.txt
# where a lock-step's time goes in the FREE-RUNNING loop (stamp kernels on the actors' stream; with SRLX_BACKWARD_STAMPS=1 also inside the update's graph), as bench.py runs it
SRLX_ACTOR_STREAM=low python tools/freerun_phases.py 2>&1 | grep -v amdgpu | tail -6 > gpurun_out/r6_freerun_phases.txt; cat gpurun_out/r6_freerun_phases.txt
SRLX_ACTOR_STREAM=low SRLX_BACKWARD_STAMPS=1 python tools/freerun_phases.py 2>&1 | grep "free-running" > gpurun_out/r6_freerun_update_phases.txt; cat gpurun_out/r6_freerun_update_phases.txt
bash tools/_trace_learner_fast.sh > gpurun_out/r6_learner_alone_timeline.txt 2>&1; head -2 gpurun_out/r6_learner_alone_timeline.txt
bash tools/_pmc_conv.sh 2>&1 | grep "^conv\|^fc1\|^head" > gpurun_out/r6_pmc_sq_counters.txt; head -2 gpurun_out/r6_pmc_sq_counters.txt
bash tools/_pmc_per.sh > gpurun_out/r6_per_pmc.log 2>&1; tail -8 gpurun_out/r6_per_pmc.log
bash tools/_roles.sh  # the multi-GPU job's roles, each alone on this GPU, in processes of their own: r6_roles_n8.json (7 actor ranks), r6_roles_n4.json (3), r6_roles_n2_rank0.json
python tools/_shim.py 2>/dev/null | grep -v amdgpu > gpurun_out/r6_shim_speedtest.json; grep "_us" gpurun_out/r6_shim_speedtest.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profa
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profa -- python $R/bench.py --algo agent57_light --envs 1024 --capacity 200000 --steps 4 --inner 16 --warmup 1 > /dev/null 2>&1
f=$(find /tmp/profa -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $R/gpurun_out/r6_a57_kernel_stats.csv
cd $R
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profp -- python $R/bench.py --algo ppo --steps 30 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/profp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" > $R/gpurun_out/r6_ppo_kernel_stats.csv
cd $R
