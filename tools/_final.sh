cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_gputest.log; tail -3 gpurun_out/r6_gputest.log
