cd $GRAFT_REPO_ROOT
timeout 300 python bench.py --algo ppo --steps 30 > gpurun_out/r6_bench_ppo.json 2>/dev/null; python -c "import json;d=json.loads(open('gpurun_out/r6_bench_ppo.json').read().strip().splitlines()[-1]);print('ppo', round(d['value']), d['ms_per_step'], d['learner_updates_per_s'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['roofline']['share_of_iteration'])"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profp -- python $GRAFT_REPO_ROOT/bench.py --algo ppo --steps 30 --no-cpu-baseline > /dev/null 2>&1
f=$(find /tmp/profp -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -20 "$f" > $GRAFT_REPO_ROOT/gpurun_out/r6_ppo_kernel_stats.csv; head -6 $GRAFT_REPO_ROOT/gpurun_out/r6_ppo_kernel_stats.csv | cut -c1-150
