# Round-end measurement pass on the GPU box: full GPU suite, smoke, bench (+ PER micro), rocprofv3 kernel stats of the
# same bench command, PER bulk-sampling probe with its kernel stats, and the N>1 code path at world size 1.
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --per-micro > gpurun_out/bench_r1.json 2> gpurun_out/bench_r1.err; tail -c 1500 gpurun_out/bench_r1.json
python bench.py --dist-selftest --no-cpu-baseline 2>/dev/null | grep metric > gpurun_out/bench_r1_dist_selftest.json
python tools/per_probe.py > gpurun_out/per_probe.log 2>&1; cat gpurun_out/per_probe.log | cut -c1-150
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profb -- python $R/bench.py --steps 200 --warmup 20 --cpu-seconds 0 > $R/gpurun_out/bench_r1_profiled.json 2>/dev/null
f=$(find /tmp/profb -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r1_kernel_stats.csv
python - "$f" $R/gpurun_out/bench_r1_profiled.json > $R/gpurun_out/r1_roofline_check.txt <<'PY'
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = json.loads([l for l in open(sys.argv[2]) if l.startswith('{')][0])
print("same command, same run (python bench.py --steps 200 --warmup 20 --cpu-seconds 0 under rocprofv3 --kernel-trace --stats):")
for r in rows:
    if 'AConv, 64, true, false, 128' in r['Name']:
        print("rocprofv3 kernel_stats: k_gemm<AConv, 64, true, false, 128>  calls %s  AverageNs %s  -> two launches per step = %.1f us" % (r['Calls'], r['AverageNs'], 2 * float(r['AverageNs']) / 1e3))
print("bench.py roofline (HIP events on the launch stream, around the two launches): avg_launch_pair_ms = %.4f ms (includes the gap between the two launches)" % d['roofline']['avg_launch_pair_ms'])
print("bench.py of that run: ms_per_step %.4f (kernel tracing serialises the streams; the unprofiled run is profiles/r1_bench.json)" % d['ms_per_step'])
PY
cat $R/gpurun_out/r1_roofline_check.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profp -- python $R/tools/per_probe.py quick > /dev/null 2>&1
f=$(find /tmp/profp -name "*kernel_stats.csv" | head -1); grep -v "at::native" "$f" | cut -c1-400 > $R/gpurun_out/r1_per_kernel_stats.csv
head -8 $R/gpurun_out/r1_per_kernel_stats.csv | cut -c1-200
