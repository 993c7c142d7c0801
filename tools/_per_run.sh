#!/bin/bash
# PER parity tests, then the bulk-sampling probe under rocprofv3 (per-kernel times)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_per_gpu.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/per_tests.log
timeout 300 python tools/per_probe.py > gpurun_out/per_probe.log 2>&1
export TMPDIR=/tmp
rm -rf gpurun_out/prof_per
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_per -- python tools/per_probe.py > /dev/null 2>&1
f=$(find gpurun_out/prof_per -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-200 | grep -v "at::native" > gpurun_out/per_kernel_stats.csv
find gpurun_out/prof_per -type f ! -name '*stats.csv' -delete
cat gpurun_out/per_tests.log gpurun_out/per_probe.log gpurun_out/per_kernel_stats.csv
