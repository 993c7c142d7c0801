cd $GRAFT_REPO_ROOT; timeout 300 python tools/per_add_probe.py; timeout 900 python -m pytest tests/test_per_gpu.py -x -q -m gpu 2>&1 | tail -2
