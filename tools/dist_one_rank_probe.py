"""Rank 0 of the 2-GPU topology ALONE on one GPU: `DistributedRainbow` at world size 1 with `learner_acts` -- the single-GPU lock-step on its own E environments whose
update ingests the slab through the packed commit (`put_own` into the staging slot, `srlx_store_commit_step_packed` + tree add on the update's side branch), over a
one-rank RCCL group.  Prints ms per lock-step (bench.py's N = 1 line runs `RainbowEngine` directly: no packed commit, no staging copy)."""
import json, os, socket, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "2")
import torch
import torch.distributed as dist
from simple_distributed_rl_amd.device.dist import DistributedRainbow
from simple_distributed_rl_amd.device.rainbow import RainbowDeviceConfig

E = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda:0"))
cfg = RainbowDeviceConfig(n_envs=E, batch_size=32, memory_capacity=1_000_000, seed=0)
job = DistributedRainbow(cfg, 0, episode_len=200, sync_interval=16, learner_acts=True, actor_stream="low")
job.prefill()
for _ in range(16):
    job.step(1)
job.capture_graphs()
for _ in range(64):
    job.step(1)
torch.cuda.synchronize()
n = 512
t0 = time.perf_counter()
for _ in range(n):
    job.step(1)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(json.dumps({"ms_per_lock_step": 1e3 * dt, "env_steps_per_s": E / dt, "envs": E, "fast": bool(job.local.fast), "role": job.local.role, "info": job.info()}))
job.local.close()
dist.destroy_process_group()
