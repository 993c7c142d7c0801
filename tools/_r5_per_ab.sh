#!/bin/bash
# round 5: bulk PER, SAME-box A/B of libsrlx builds under tools/_abl/ (arguments: build names), parity tests of the current build first
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_per_gpu.py -x -q -m gpu 2>&1 | tail -3
for rep in 1 2 3; do
for k in "$@"; do
  echo "== $k"
  SRLX_LIB=$PWD/tools/_abl/libsrlx_$k.so python tools/per_probe.py quick 2>&1 | grep "draws" | cut -c1-150
done; done
