"""`Runner.train_mp()` on the device engine: one process per GPU over RCCL.

The reference's `train_mp(actor_num, actor_devices, ...)` (srl/runner/runner.py:310-402 -> srl/base/run/play_mp.py:471-642)
spawns `actor_num` actor processes and one trainer process that talk through a pickling queue and a parameter board.
Here the same call maps the roles onto GPUs:

    the CALLING process is the learner rank (rank 0): it owns the global replay (frame ring + sum-tree), runs the
        updates, fires the trainer-side callbacks and ends with the trained weights in `runner.parameter`;
    actor i runs on `actor_devices[i]` as one spawned process with E lock-stepped environments;
    an actor whose device is the learner's device is hosted BY the learner rank (rank 0 acts and learns);
    transitions travel as fixed-size slabs (gather), weights as one flat broadcast every `sync_interval_steps`
        lock-steps (device/dist.py), instead of pickled items and a pickled state_dict on a timer.

Backend: "nccl" (= RCCL on ROCm) when every rank has its own GPU; "gloo" when ranks share a GPU (single-GPU test
rigs: tensors are staged through the host, everything else is the same code).

Trainer-side hook protocol of play_mp.py:321-468 (`on_trainer_start`, `on_train_before`, `on_train_after`,
`on_trainer_end` with a RunStateTrainer): before/after fire once per lock-step; `state.train_count`,
`state.trainer_recv_q` (transitions received) and `state.sync_trainer` (broadcasts sent) advance accordingly.
Stop rules: `max_train_count` and `timeout` (the trainer's rules in the reference), decided by the learner rank and
told to the actor ranks with the flag all-reduce that every rank joins each `check_every` lock-steps.
"""
import os
import socket
import time
from typing import List, Optional, Sequence

import torch

from simple_distributed_rl_amd.base.context import RunContext, RunStateTrainer
from simple_distributed_rl_amd.base.run.hooks import HookTable


def _device_index(name: str, default: int = 0) -> int:
    name = str(name).lower()
    if ":" in name:
        return int(name.split(":")[1])
    return default


def plan_ranks(learner_device: str, actor_num: int, actor_devices, memory_device=None) -> dict:
    """Which GPU every rank drives.  rank 0 = learner.  Returns dict(devices=[gpu index per rank], learner_acts, backend).
    memory_device: the reference's THREE-role topology (play_mp_memory.py) -- rank 1 is a replay GPU, ranks 2.. act, nobody else does."""
    n_gpu = max(1, torch.cuda.device_count())
    if memory_device is not None:
        if isinstance(actor_devices, str):
            used = {_device_index(learner_device), _device_index(memory_device)}
            free = [g for g in range(n_gpu) if g not in used] or [_device_index(learner_device)]
            actor_devices = [f"cuda:{free[i % len(free)]}" for i in range(actor_num)] if actor_devices.upper() in ("AUTO", "GPU", "CUDA") else [actor_devices] * actor_num
        ranks = [_device_index(learner_device), _device_index(memory_device)] + [_device_index(d) for d in actor_devices]
        assert len(ranks) == actor_num + 2, "one device per actor"
        return dict(devices=ranks, learner_acts=False, backend="nccl" if len(set(ranks)) == len(ranks) else "gloo", replay_role=True)
    if isinstance(actor_devices, str):
        if actor_devices.upper() in ("AUTO", "GPU", "CUDA"):
            # spread the actors over the GPUs that are not the learner's; with a single GPU everybody shares it
            free = [g for g in range(n_gpu) if g != _device_index(learner_device)] or [_device_index(learner_device)]
            actor_devices = [f"cuda:{free[i % len(free)]}" for i in range(actor_num)]
        else:
            actor_devices = [actor_devices] * actor_num
    devs = [_device_index(d) for d in actor_devices]
    assert len(devs) == actor_num, "one device per actor"
    learner = _device_index(learner_device)
    learner_acts = learner in devs
    if learner_acts:
        devs.remove(learner)  # that actor lives inside the learner rank
    ranks = [learner] + devs
    backend = "nccl" if len(set(ranks)) == len(ranks) else "gloo"
    return dict(devices=ranks, learner_acts=learner_acts, backend=backend)


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _init_group(rank: int, world: int, port: int, backend: str, device: int):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.cuda.set_device(device)
    if backend == "nccl":
        from simple_distributed_rl_amd.device.dist import rccl_options

        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{device}"), pg_options=rccl_options())
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


class _JobLoop:
    """The lock-step loop every rank runs; only the learner rank decides when to stop."""

    def __init__(self, engine, backend: str, updates_per_step: int, check_every: int):
        import torch.distributed as dist

        self.eng, self.dist, self.backend = engine, dist, backend
        self.updates, self.check_every = int(updates_per_step), max(1, int(check_every))
        self._flag = torch.zeros(1, dtype=torch.int32, device=engine.dev if backend == "nccl" else "cpu")

    def agree_to_stop(self, want: bool) -> bool:
        self._flag.fill_(1 if want else 0)
        self.dist.all_reduce(self._flag, op=self.dist.ReduceOp.MAX)
        return bool(self._flag.item())

    def run(self, should_stop, before=None, after=None):
        eng = self.eng
        k = 0
        while True:
            if k % self.check_every == 0 and self.agree_to_stop(should_stop()):
                break
            if before is not None:
                before()
            eng.step(self.updates)
            if after is not None and after():
                should_stop = lambda: True  # noqa: E731  (a callback asked to stop: tell everybody at the next check)
            k += 1
        eng.flush()


def _make_engine(kind: str, cfg, device: int, plan: dict, env_spec: dict, opts: dict, parameter=None):
    """The rank's distributed engine: DistributedRainbow (cfg = RainbowDeviceConfig) or DistributedAgent57Light (cfg = the set-up rl_config)."""
    from simple_distributed_rl_amd.device import dist as D

    if plan.get("replay_role"):
        from simple_distributed_rl_amd.device.replay_role import ReplayRoleRainbow

        assert kind == "rainbow", "the replay-GPU topology serves the Rainbow family"
        return ReplayRoleRainbow(cfg, device, sync_interval=opts["sync_interval_steps"], prefetch=opts.get("prefetch", 5), updates=opts["updates_per_step"], env=_env_factory(env_spec))
    if kind == "agent57_light":
        if not cfg.is_setup():
            from simple_distributed_rl_amd.base.env.registration import make as make_env_run

            cfg.setup(make_env_run(env_spec["env_config"]))
        from simple_distributed_rl_amd.device.agent57_fast import why_not_fast

        cls = D.DistributedAgent57Light if not why_not_fast(cfg) else D.DistributedAgent57LightGeneral  # (84 x 84 x 4: all-libsrlx engine + slot exchange)
        return cls(cfg, opts["lanes"], device, sync_interval=opts["sync_interval_steps"], learner_acts=plan["learner_acts"],
                   seed=int(env_spec.get("seed") or 0), env=_env_factory(env_spec), parameter=parameter)
    return D.DistributedRainbow(cfg, device, sync_interval=opts["sync_interval_steps"], learner_acts=plan["learner_acts"], env=_env_factory(env_spec))


def _actor_rank_main(rank: int, world: int, port: int, plan: dict, kind: str, cfg, env_spec: dict, opts: dict):
    """Entry point of a spawned actor rank."""
    dist = _init_group(rank, world, port, plan["backend"], plan["devices"][rank])
    try:
        eng = _make_engine(kind, cfg, plan["devices"][rank], plan, env_spec, opts)
        if plan.get("replay_role"):
            eng.broadcast_weights()
        else:
            eng.bus.broadcast_params(eng.flat)  # the weights the learner rank started from (runner.parameter)
        _JobLoop(eng, plan["backend"], opts["updates_per_step"], opts["check_every"]).run(lambda: False)
    finally:
        dist.destroy_process_group()


def _env_factory(env_spec: dict):
    """replay -> batch environment, from a picklable description of the Runner's environment."""

    def make(replay):
        from simple_distributed_rl_amd.base.env.registration import EnvConfig, make as make_env_run
        from simple_distributed_rl_amd.device.vector_runner import HostVecEnv

        env_config: EnvConfig = env_spec["env_config"]
        probe = make_env_run(env_config)
        maker = getattr(type(probe.unwrapped), "device_vector", None)
        if maker is not None:
            return maker(replay, **env_config.kwargs)
        env = HostVecEnv(env_config, replay.E, replay.dev, env_spec.get("seed"), processor=env_spec.get("processor"))
        env.setup(None)
        return env

    return make


def train_mp_on_engine(runner, context: RunContext, lanes: int, actor_num: int, actor_devices, updates_per_step: int = 1,
                       sync_interval_steps: int = 16, check_every: int = 16, memory_device=None, prefetch: int = 5,
                       actor_initial_priority: bool = True) -> RunStateTrainer:
    """memory_device (e.g. "cuda:1"): the reference's three-role topology with the replay on a GPU of its own (device/replay_role.py); default: the
    learner rank owns the replay (device/dist.py)."""
    from simple_distributed_rl_amd.device import vector_runner as vr

    context.check_context_parameter()
    plan = plan_ranks(context.used_device_torch, actor_num, actor_devices, memory_device)
    world = len(plan["devices"])
    seed = 0 if context.seed is None else int(context.seed)
    kind = vr.engine_kind(runner.rl_config)
    cfg = runner.rl_config if kind == "agent57_light" else vr.device_config_from(runner.rl_config, runner.make_env(), lanes, seed)
    if actor_initial_priority and kind == "rainbow" and not plan.get("replay_role") and runner.rl_config.memory.name != "ReplayBuffer":
        import dataclasses

        # the reference's DISTRIBUTED workers estimate an item's first priority themselves (rainbow.py:389-398, `memory.requires_priority()`): so do the actor
        # ranks (two launches per lock-step: +5 % at 1024 environments); train_mp(..., actor_initial_priority=False) keeps max_priority
        cfg = dataclasses.replace(cfg, actor_initial_priority=True)
    env_spec = dict(env_config=runner.env_config, seed=context.seed, processor=vr.frame_processor(runner.rl_config))
    opts = dict(updates_per_step=updates_per_step, sync_interval_steps=sync_interval_steps, check_every=check_every, lanes=lanes, prefetch=prefetch)
    port = _free_port()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = None
    if world > 1:
        procs = _spawn_ranks(world, port, plan, kind, cfg, env_spec, opts)
    state = RunStateTrainer()
    hooks = HookTable(context.callbacks, context=context, state=state)
    dist = _init_group(0, world, port, plan["backend"], plan["devices"][0])
    eng = None
    try:
        parameter = runner.make_parameter()
        eng = _make_engine(kind, cfg, plan["devices"][0], plan, env_spec, opts, parameter=parameter)  # agent57_light trains `parameter` in place
        if kind == "rainbow":
            _load_reference_weights(eng, parameter)
        if plan.get("replay_role"):
            eng.broadcast_weights()
        else:
            eng.bus.broadcast_params(eng.flat)
        state.parameter, state.memory, state.trainer = parameter, (vr._ReplayFacade(eng.replay) if eng.replay is not None else None), eng
        hooks.fire("on_start")
        hooks.fire("on_trainer_start")
        state.elapsed_t0 = time.time()
        trained = (lambda: eng.local.train_count) if kind == "rainbow" else (lambda: eng.train_count)
        start_count = trained()
        E_total = eng.global_envs

        def should_stop() -> bool:
            if context.timeout > 0 and time.time() - state.elapsed_t0 >= context.timeout:
                state.end_reason = "timeout."
                return True
            if context.max_train_count > 0 and state.train_count >= context.max_train_count:
                state.end_reason = "max_train_count over."
                return True
            return False

        def after() -> bool:
            done = trained() - start_count
            state.is_step_trained = done > state.train_count
            state.train_count = done
            state.trainer_recv_q += E_total
            state.sync_trainer = eng.step_count // eng.sync_interval
            if hooks.poll("on_train_after"):
                state.end_reason = "callback.trainer_intermediate_stop"
                return True
            return False

        _JobLoop(eng, plan["backend"], updates_per_step, check_every).run(should_stop, before=lambda: hooks.fire("on_train_before"), after=after)
        if kind == "rainbow":
            _store_reference_weights(eng, parameter)
        elif hasattr(eng.local, "export_parameter"):  # the all-libsrlx Agent57_light engine trains masters of its own
            eng.local.export_parameter(parameter)
        state.shared_vars["actor_env_steps"] = eng.step_count * E_total
    finally:
        try:
            hooks.fire("on_trainer_end")
            hooks.fire("on_end")
        finally:
            dist.destroy_process_group()
            if procs is not None:
                _join_ranks(procs)
    return state


def _spawn_ranks(world, port, plan, kind, cfg, env_spec, opts):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_actor_rank_main, args=(r, world, port, plan, kind, cfg, env_spec, opts), daemon=True) for r in range(1, world)]
    for p in procs:
        p.start()
    return procs


def _join_ranks(procs):
    bad = []
    for p in procs:
        p.join(timeout=120)
        if p.is_alive():
            p.terminate()
            bad.append("hung")
        elif p.exitcode != 0:
            bad.append(p.exitcode)
    if bad:
        raise RuntimeError(f"actor rank(s) ended with {bad}")  # play_mp.py:623-635


def _load_reference_weights(eng, parameter):
    local = eng.local
    online, target = parameter.q_online.state_dict(), parameter.q_target.state_dict()
    nets = [(local.q_online, online), (local.q_target, target)]
    if local.q_actor is not local.q_online:
        nets.append((local.q_actor, online))
    for net, sd in nets:
        (net.load_reference_state_dict if hasattr(net, "load_reference_state_dict") else net.load_state_dict)(sd)


def _store_reference_weights(eng, parameter):
    local = eng.local
    torch.cuda.synchronize(eng.dev)
    for mine, theirs in ((local.q_online, parameter.q_online), (local.q_target, parameter.q_target)):
        sd = mine.reference_state_dict() if hasattr(mine, "reference_state_dict") else mine.state_dict()
        theirs.load_state_dict({k: v.to(next(theirs.parameters()).device) for k, v in sd.items()})
