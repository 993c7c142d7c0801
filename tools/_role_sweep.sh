#!/bin/bash
# The two roles alone at E environments per actor GPU (7 actor ranks) + the same E_total on ONE GPU: what the choice of E_total does to the predicted strong ratio
cd ${GRAFT_REPO_ROOT:-.}
for E in 768 1024 1280 1536; do
  python tools/role_probe.py --envs $E 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); a=d['actor_rank']; l=d['learner_rank']; p=d['predicted']
print('E per actor GPU %d: actor rank %.4f ms, learner rank %.4f ms per period (slab %d), predicted %.2f M env-steps/s at 8 GPUs' % (a['envs'], a['ms_per_lock_step'], l['ms_per_period'], l['slab_envs'], p['env_steps_per_s']/1e6))"
  SRLX_NO_ROLES=1 python bench.py --envs $((7*E)) --steps 6 --warmup 2 --inner 32 --no-cpu-baseline --no-per-micro --no-subfigures 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('   one GPU, %d environments: %.3f M env-steps/s, %.4f ms per lock-step, %.0f updates/s' % (d['config']['envs_total'], d['value']/1e6, d['ms_per_lock_step'], d['learner_updates_per_s']))"
done
