cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_engine_gpu.py -x -q -m gpu -k cue 2>&1 | tail -15
