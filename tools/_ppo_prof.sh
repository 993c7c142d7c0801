cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/pp -o p --output-format csv -- python bench.py --algo ppo --steps 30 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
f=$(find gpurun_out/pp -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/r6_ppo_kernel_stats.csv; rm -rf gpurun_out/pp; head -8 gpurun_out/r6_ppo_kernel_stats.csv | cut -c1-160
