#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
one() { env "$@" timeout 600 python bench.py --algo agent57_light --steps 6 --inner 16 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-50s %9.0f env-steps/s %.3f ms per lock-step' % ('$*', d['value'], d['ms_per_lock_step']))"; }
{
one A=base
one TORCH_BLAS_PREFER_HIPBLASLT=0
one PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=30 PYTORCH_TUNABLEOP_FILENAME=/tmp/tunable.csv
} 2>&1 | tee gpurun_out/r4_a57b.log
