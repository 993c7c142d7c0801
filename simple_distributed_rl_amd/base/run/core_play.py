"""The sequential actor+learner loop (srl/base/run/core_play.py:15-238): one iteration = one env step;
stop checks, episode reset, policy -> env.step -> on_step, `train_repeat` trainer calls every
`train_interval` steps, callback hooks at the reference's points, teardown in `finally`."""
import logging
import random
import time
from typing import List, Optional

from simple_distributed_rl_amd.base.context import RunContext, RunStateActor
from simple_distributed_rl_amd.utils import common

logger = logging.getLogger(__name__)


def play(context: RunContext, env, worker, trainer=None, workers: Optional[List] = None, state: Optional[RunStateActor] = None):
    context.check_context_parameter()
    context.setup_device()
    callbacks = context.callbacks

    if state is None:
        state = RunStateActor()
    state.env = env
    state.worker = worker
    state.parameter = worker.worker.parameter
    state.memory = worker.worker.memory
    if workers is None:
        workers, _ = context.rl_config.make_workers(context.players, env, state.parameter, state.memory, worker)
    state.workers = workers
    if context.disable_trainer:
        trainer = None
    elif context.training and trainer is None:
        trainer = context.rl_config.make_trainer(state.parameter, state.memory)
    state.trainer = trainer
    assert env.player_num == len(workers)

    if not context.distributed:
        [c.on_start(context=context, state=state) for c in callbacks]
    if context.distributed:
        worker.config.setup_from_actor(context.actor_num, context.actor_id)
    if context.seed is not None:
        common.set_seed(context.seed, context.seed_enable_gpu)
        state.episode_seed = random.randint(0, 2 ** (16 - 4))  # core_play.py:79

    env.setup(context)
    [w.setup(context, run_state=state) for w in workers]
    if trainer is not None:
        trainer.setup(context)
    state.worker_indices = list(range(env.player_num))

    def hooks(name):
        return [c for c in callbacks if hasattr(c, name)]

    h_ep_begin, h_ep_end = hooks("on_episode_begin"), hooks("on_episode_end")
    h_act_before, h_act_after = hooks("on_step_action_before"), hooks("on_step_action_after")
    h_step_begin, h_step_end = hooks("on_step_begin"), hooks("on_step_end")
    [c.on_episodes_begin(context=context, state=state) for c in callbacks]

    try:
        state.elapsed_t0 = time.time()
        while True:
            if context.timeout > 0 and (time.time() - state.elapsed_t0) >= context.timeout:
                state.end_reason = "timeout."
                break
            if context.max_steps > 0 and state.total_step >= context.max_steps:
                state.end_reason = "max_steps over."
                break
            if trainer is not None and context.max_train_count > 0 and state.train_count >= context.max_train_count:
                state.end_reason = "max_train_count over."
                break
            if state.memory is not None and context.max_memory > 0 and state.memory.length() >= context.max_memory:
                state.end_reason = "max_memory over."
                break

            if env.done:
                state.episode_count += 1
                if context.max_episodes > 0 and state.episode_count >= context.max_episodes:
                    state.end_reason = "episode_count over."
                    break
                env.reset(seed=state.episode_seed)
                if state.episode_seed is not None:
                    state.episode_seed += 1
                if context.shuffle_player:
                    random.shuffle(state.worker_indices)
                state.worker_idx = state.worker_indices[env.next_player]
                [w.reset(state.worker_indices[i], seed=state.episode_seed) for i, w in enumerate(workers)]
                [c.on_episode_begin(context=context, state=state) for c in h_ep_begin]

            [c.on_step_begin(context=context, state=state) for c in h_step_begin]
            [c.on_step_action_before(context=context, state=state) for c in h_act_before]
            state.action = workers[state.worker_idx].policy()
            [c.on_step_action_after(context=context, state=state) for c in h_act_after]

            if not env.done:
                env.step(state.action, workers[state.worker_idx].config.frameskip)
                [w.on_step() for w in workers]
                state.total_step += 1

            if trainer is not None and state.total_step % context.train_interval == 0:
                prev = trainer.train_count
                for _ in range(context.train_repeat):
                    trainer.train()
                state.is_step_trained = trainer.train_count > prev
                if state.is_step_trained:
                    state.train_count += trainer.train_count - prev

            stop_flags = [c.on_step_end(context=context, state=state) for c in h_step_end]
            state.worker_idx = state.worker_indices[env.next_player]

            if env.done:
                rewards = [env.episode_rewards[state.worker_indices[i]] for i in range(env.player_num)]
                state.episode_rewards_list.append(rewards)
                state.last_episode_step = env.step_num
                state.last_episode_time = env.elapsed_time
                state.last_episode_rewards = rewards
                [c.on_episode_end(context=context, state=state) for c in h_ep_end]
            if True in stop_flags:
                state.end_reason = "callback.intermediate_stop"
                break
    finally:
        env.teardown()
        [w.teardown() for w in workers]
        if trainer is not None:
            trainer.teardown()
        if state.episode_count == 0 and env.step_num > 0 and len(state.episode_rewards_list) == 0:
            rewards = [env.episode_rewards[state.worker_indices[i]] for i in range(env.player_num)]
            state.episode_rewards_list.append(rewards)
            state.last_episode_step = env.step_num
            state.last_episode_time = env.elapsed_time
            state.last_episode_rewards = rewards
        [c.on_episodes_end(context=context, state=state) for c in callbacks]
        if not context.distributed:
            [c.on_end(context=context, state=state) for c in callbacks]
    return state
