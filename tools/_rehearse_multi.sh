#!/bin/bash
# Rehearsal of `bench.py --gpus N` under the driver's launcher with N ranks SHARING the one GPU of the box (gloo rendezvous, tensors staged through the host):
# everything of the N > 1 path except the RCCL transport itself (whose stream semantics tests/test_dist_stream_semantics_gpu.py covers) -> gpurun_out/r6_rehearsal_nN.json
cd $GRAFT_REPO_ROOT
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --backend gloo --steps 6 --warmup 1 --inner 16 --envs 512 --capacity 200000 2>gpurun_out/rehearse_$n.err | grep '"metric"' > gpurun_out/r6_rehearsal_n$n.json
  python -c "
import json,sys
d=json.loads(open('gpurun_out/r6_rehearsal_n$n.json').read()); print(d['n_gpus'], round(d['value']), d['ms_per_step'], d['config']['topology'][:60], d['config']['actor_gpus'], d['scaling'], d['rccl_ranks'], 'strong_ref', d.get('strong_ref',{}).get('value'), 'ratio', d.get('strong_ratio'), d['final'])" || tail -20 gpurun_out/rehearse_$n.err
done
# the other two workloads' N > 1 lines (2 ranks sharing the GPU): Agent57_light (7 + 1 topology at world 2 = 1 + 1), PPO (data parallel)
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29610 bench.py --gpus 2 --backend gloo --algo agent57_light --steps 3 --warmup 1 --inner 8 --envs 512 --capacity 100000 2>gpurun_out/rehearse_a57.err | grep '"metric"' > gpurun_out/r6_rehearsal_a57_n2.json
python -c "
import json; d=json.loads(open('gpurun_out/r6_rehearsal_a57_n2.json').read()); print('a57 n2', round(d['value']), d['ms_per_lock_step'], d['config'].get('parallelism','')[:80])" || tail -20 gpurun_out/rehearse_a57.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --backend gloo --algo ppo --steps 5 --warmup 2 2>gpurun_out/rehearse_ppo.err | grep '"metric"' > gpurun_out/r6_rehearsal_ppo_n2.json
python -c "
import json; d=json.loads(open('gpurun_out/r6_rehearsal_ppo_n2.json').read()); print('ppo n2', round(d['value']), d['ms_per_step'], d['config']['parallelism'][:90], d['config']['hip_graphs'])" || tail -20 gpurun_out/rehearse_ppo.err
