#!/bin/bash
# Rehearsal of `bench.py --gpus N` under the driver's launcher with N ranks SHARING the one GPU of the box (gloo rendezvous,
# tensors staged through the host): exercises everything of the N>1 path except the RCCL transport itself.
cd $GRAFT_REPO_ROOT
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --backend gloo --steps 30 --warmup 5 --envs 256 --capacity 200000 2>gpurun_out/rehearse_$n.err | grep metric | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['ms_per_step'], d['config']['topology'], d['config']['actor_gpus'], d['final'])" || tail -20 gpurun_out/rehearse_$n.err
done
