#!/bin/bash
# same-box A/B of two builds of libsrlx on the bulk PER probe: tools/_ab/libsrlx_old.so against the in-tree library, interleaved twice
R=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for lib in $R/tools/_ab/libsrlx_old.so $R/simple_distributed_rl_amd/libsrlx.so; do
    echo "== $lib"; SRLX_LIB=$lib python $R/tools/per_probe.py quick 2>&1 | tail -2
  done
done
