mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --per-micro > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err; tail -c 1500 gpurun_out/bench_r1.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -- python $GRAFT_REPO_ROOT/bench.py --steps 200 --warmup 20 --cpu-seconds 0 > $GRAFT_REPO_ROOT/gpurun_out/bench_r1_profiled.json 2>/dev/null
f=$(find /tmp/profb -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r1_kernel_stats.csv
