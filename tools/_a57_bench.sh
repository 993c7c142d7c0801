cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_agent57_fast_gpu.py tests/test_dist_gpu.py tests/test_dist_stream_semantics_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 500 python bench.py --algo agent57_light --envs 1024 --capacity 200000 --steps 4 --inner 16 --warmup 1 > gpurun_out/r6_bench_agent57_light.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r6_bench_agent57_light.json').read().strip().splitlines()[-1]);print('agent57_light', round(d['value']), d['ms_per_lock_step'], d['learner_updates_per_s'], d['roofline']['frac'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profa
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profa -- python $GRAFT_REPO_ROOT/bench.py --algo agent57_light --envs 1024 --capacity 200000 --steps 4 --inner 16 --warmup 1 > /dev/null 2>&1
f=$(find /tmp/profa -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r6_a57_kernel_stats.csv
cd $GRAFT_REPO_ROOT; python tools/graph_replay_check.py 2>&1 | grep "agent57_light: distinct"
