#!/bin/bash
# round 5: what the bulk PER walk spends its time on -- libsrlx builds with one piece of k_descend_bulk removed each (tools/_abl/, -DSRLX_BULK_ABL=k), interleaved on one box
cd $GRAFT_REPO_ROOT
tools/_gather_probe 17 20; tools/_gather_probe 17 22; tools/_gather_probe 14 22
for rep in 1 2; do
for k in 0 1 2 3 4 5; do
  echo "== abl $k (0 shipped, 1 no IS-weight math, 2 no stores, 3 group fetches from 64 KB, 4 stage 293 blocks only, 5 walk launch only)"
  SRLX_LIB=$PWD/tools/_abl/libsrlx_abl$k.so python tools/per_probe.py quick 2>&1 | grep "zeros 0.0 draws" | cut -c1-110
done; done
