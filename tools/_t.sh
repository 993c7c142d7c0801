cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_dist_gpu.py -x -q -m gpu 2>&1 | tail -5
timeout 900 python bench.py --dist-selftest --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | grep metric | cut -c1-330
