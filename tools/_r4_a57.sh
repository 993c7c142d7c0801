#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
{
echo "--- bench agent57_light E=1024 (untraced)"
timeout 600 python bench.py --algo agent57_light --steps 6 --inner 16 --warmup 1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_lock_step'])"
echo "--- actor only / learner only timings"
timeout 600 python - <<'PY'
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import simple_distributed_rl_amd as srl
from simple_distributed_rl_amd.algorithms import agent57_light
from simple_distributed_rl_amd.device.agent57_light import Agent57LightEngine
rl = agent57_light.Config(batch_size=32); rl.window_length = 4
rl.memory.capacity, rl.memory.warmup_size = 200_000, 80_000
rl.memory.set_proportional(alpha=0.6, beta_initial=0.4, beta_steps=1_000_000)
rl.input_block.image.set_dqn_block(); rl.hidden_block.set_dueling_network((512,))
env = srl.make_env(srl.EnvConfig("SyntheticAtari-v0", kwargs=dict(episode_len=200))); rl.setup(env)
eng = Agent57LightEngine(rl, 1024, 0, episode_len=200, seed=0); eng.prefill()
for _ in range(16): eng.step(1)
torch.cuda.synchronize(); eng.capture_graphs()
def timed(fn, n=48):
    for _ in range(4): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    th=time.perf_counter()-t; torch.cuda.synchronize(); return 1e3*th/n, 1e3*(time.perf_counter()-t)/n
print("actor_step   host %.3f ms  wall %.3f ms" % timed(eng.actor_step))
print("actor_net    host %.3f ms  wall %.3f ms" % timed(lambda: eng.actor_net()))
print("learner_step host %.3f ms  wall %.3f ms" % timed(eng.learner_step))
print("step(1)      host %.3f ms  wall %.3f ms" % timed(lambda: eng.step(1)))
PY
echo "--- kernel stats E=1024"
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/profa
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profa -- python $R/bench.py --algo agent57_light --steps 4 --inner 16 --warmup 1 > /tmp/a57.json 2>/dev/null
python $R/tools/kstats.py /tmp/profa 45
f=$(find /tmp/profa -name "*kernel_stats.csv" | head -1); cp "$f" $R/gpurun_out/r4_a57_kernel_stats_before.csv
} 2>&1 | tee gpurun_out/r4_a57.log
