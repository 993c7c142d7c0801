#!/bin/bash
# Rehearsal of `bench.py --gpus N` under the driver's launcher with N ranks SHARING the one GPU of the box (gloo rendezvous, tensors staged through the host):
# everything of the N > 1 path except the RCCL transport itself (whose stream semantics tests/test_dist_stream_semantics_gpu.py covers) -> gpurun_out/r5_rehearsal_nN.json
cd $GRAFT_REPO_ROOT
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --backend gloo --steps 6 --warmup 1 --inner 16 --envs 512 --capacity 200000 2>gpurun_out/rehearse_$n.err | grep '"metric"' > gpurun_out/r5_rehearsal_n$n.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r5_rehearsal_n$n.json').read()); print(d['n_gpus'], round(d['value']), d['ms_per_step'], d['config']['topology'][:60], d['config']['actor_gpus'], d['scaling'], d['rccl_ranks'], 'strong_ref', d.get('strong_ref',{}).get('value'), 'ratio', d.get('strong_ratio'), d['final'])" || tail -20 gpurun_out/rehearse_$n.err
done
