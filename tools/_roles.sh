mkdir -p gpurun_out
python bench.py --roles-only 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r6_roles_n8.json
python bench.py --roles-only --actor-ranks 3 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r6_roles_n4.json
python tools/dist_one_rank_probe.py 2>/dev/null | grep '^{' | tail -1 > gpurun_out/r6_roles_n2_rank0.json
python - <<'PY'
import json
for f in ("r6_roles_n8","r6_roles_n4"):
    d=json.load(open("gpurun_out/%s.json"%f)); l=d["learner_rank"]
    print(f, "actor %.4f | learner bare %.4f fabric %.4f (%.3f) | predicted %.2f M, with fabric %.2f M"%(d["actor_rank"]["ms_per_lock_step"], l["ms_per_period"], l["fabric_ms_per_period"], l["fabric_over_bare"], d["predicted"]["env_steps_per_s"]/1e6, d["predicted"]["with_fabric"]["env_steps_per_s"]/1e6))
d=json.load(open("gpurun_out/r6_roles_n2_rank0.json")); print("n2 rank0", d["ms_per_lock_step"], d["env_steps_per_s"])
PY
