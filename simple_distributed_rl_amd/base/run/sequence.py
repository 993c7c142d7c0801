"""The play loops, built around *drivers*.

The reference has one loop body per topology with the environment, the workers and the trainer inlined
(srl/base/run/core_play.py:15-238 for actor+learner, core_train_only.py:12-98 for the learner alone); one
iteration is one step of ONE environment.  Here the loop only knows two roles:

    ActorDriver    advances `lanes` environments by one lock-step per iteration
    LearnerDriver  owes `train_repeat` updates for every `train_interval` environment steps taken

and two implementations of each exist: the plugin drivers below (one host environment, the registered
`RLWorker`/`RLTrainer` plugin classes: lanes = 1, which is exactly the reference's behaviour) and the device
drivers of `simple_distributed_rl_amd.device.vector_runner` (E environments resident in HBM, the hand-written
engine: lanes = E).  `Runner.train()` picks the pair; the loop, its stop rules and its callback protocol are
the same for both, so a `RunCallback` written against the reference sees the hooks of
srl/base/run/callback.py:11-78 in the reference's order:

    on_start, on_episodes_begin,
      { [on_episode_begin]  on_step_begin  on_step_action_before  <policy>  on_step_action_after
        <env step>  <training owed>  on_step_end  [on_episode_end] }*
    on_episodes_end, on_end

With lanes > 1 the step hooks fire once per lock-step (`state.total_step` advances by `lanes`,
`state.action` holds all lanes' actions) and the episode hooks once per finished episode.
"""
import random
import time
from typing import List, Optional

from simple_distributed_rl_amd.base.context import RunContext, RunStateActor, RunStateTrainer
from simple_distributed_rl_amd.base.run.hooks import HookTable
from simple_distributed_rl_amd.utils import common


# ---------------------------------------------------------------------------------------------
# stop rules (core_play.py:117-133, core_train_only.py:63-70)
# ---------------------------------------------------------------------------------------------
class StopRules:
    """The run ends when the first configured budget is spent.  Budgets that are zero are off."""

    def __init__(self, context: RunContext, counts_training: bool, memory=None):
        self._deadline = context.timeout if context.timeout > 0 else None
        self._steps = context.max_steps if context.max_steps > 0 else None
        self._trains = context.max_train_count if (counts_training and context.max_train_count > 0) else None
        self._items = context.max_memory if (memory is not None and context.max_memory > 0) else None
        self._memory = memory

    def reason(self, state) -> str:
        if self._deadline is not None and time.time() - state.elapsed_t0 >= self._deadline:
            return "timeout."
        if self._steps is not None and state.total_step >= self._steps:
            return "max_steps over."
        if self._trains is not None and state.train_count >= self._trains:
            return "max_train_count over."
        if self._items is not None and self._memory.length() >= self._items:
            return "max_memory over."
        return ""


# ---------------------------------------------------------------------------------------------
# driver contracts
# ---------------------------------------------------------------------------------------------
class ActorDriver:
    """Owns the environments and whatever chooses their actions."""

    lanes: int = 1

    def attach(self, context: RunContext, state: RunStateActor) -> None:
        """Fill `state.env / worker / workers / parameter / memory` -- BEFORE on_start fires: the reference populates the run state first
        (core_play.py:49-69), and callbacks read it in on_start (e.g. `state.env.get_render_interval()`).  Optional: a driver without
        run-state objects of its own leaves it empty."""

    def open(self, context: RunContext, state: RunStateActor) -> None:
        """After on_start: seeding and the setup() calls (core_play.py:71-90)."""
        raise NotImplementedError

    def roll_episodes(self, context: RunContext, state: RunStateActor, hooks: HookTable) -> bool:
        """Start new episodes where the previous ones ended.  False = the episode budget is spent."""
        raise NotImplementedError

    def act(self, context: RunContext, state: RunStateActor, hooks: HookTable) -> None:
        """Choose actions (then fire on_step_action_after), step the environments, feed the transition to
        the replay; adds the number of environment steps taken to `state.total_step`."""
        raise NotImplementedError

    def settle(self, context: RunContext, state: RunStateActor, hooks: HookTable) -> None:
        """After on_step_end: account for the episodes that ended in this lock-step (on_episode_end)."""
        raise NotImplementedError

    def close(self, context: RunContext, state: RunStateActor) -> None:
        raise NotImplementedError


class LearnerDriver:
    def attach(self, context: RunContext, state) -> None:
        """Fill `state.trainer` (and memory / parameter when no actor did) before on_start.  Optional."""

    def open(self, context: RunContext, state) -> None:
        raise NotImplementedError

    def update(self, count: int, state) -> int:
        """Attempt `count` updates; returns how many really happened (an attempt below the memory's warm-up
        does nothing: srl/base/rl/trainer.py contract, core_play.py:188-194)."""
        raise NotImplementedError

    def close(self, context: RunContext, state) -> None:
        raise NotImplementedError


class _TrainingDebt:
    """`train_repeat` updates are owed each time the step counter crosses a multiple of `train_interval`
    (core_play.py:187-194 evaluates `total_step % train_interval == 0` once per single step; with `lanes` steps per
    iteration the same updates are owed per crossed multiple)."""

    def __init__(self, interval: int, repeat: int):
        self.interval, self.repeat = max(1, int(interval)), max(0, int(repeat))

    def owed(self, steps_before: int, steps_after: int) -> int:
        return (steps_after // self.interval - steps_before // self.interval) * self.repeat


# ---------------------------------------------------------------------------------------------
# actor + learner
# ---------------------------------------------------------------------------------------------
def run_sequence(context: RunContext, actor: ActorDriver, learner: Optional[LearnerDriver] = None, state: Optional[RunStateActor] = None) -> RunStateActor:
    context.check_context_parameter()
    context.setup_device()
    if state is None:
        state = RunStateActor()
    hooks = HookTable(context.callbacks, context=context, state=state)
    top_level = not context.distributed  # inside train_mp the launcher owns on_start / on_end
    # the run state is complete when on_start fires (core_play.py:49-69 before :71-72)
    actor.attach(context, state)
    if learner is not None:
        learner.attach(context, state)
    if top_level:
        hooks.fire("on_start")
    opened_actor = opened_learner = False
    try:
        actor.open(context, state)
        opened_actor = True
        if learner is not None:
            learner.open(context, state)
            opened_learner = True
        rules = StopRules(context, learner is not None, state.memory)
        debt = _TrainingDebt(context.train_interval, context.train_repeat)
        hooks.fire("on_episodes_begin")
        state.elapsed_t0 = time.time()
        while True:
            why = rules.reason(state)
            if why:
                state.end_reason = why
                break
            if not actor.roll_episodes(context, state, hooks):
                state.end_reason = "episode_count over."
                break
            hooks.fire("on_step_begin")
            hooks.fire("on_step_action_before")
            steps_before = state.total_step
            actor.act(context, state, hooks)
            if learner is not None:
                owed = debt.owed(steps_before, state.total_step)
                done = learner.update(owed, state) if owed > 0 else 0
                state.is_step_trained = done > 0
                state.train_count += done
            halt = hooks.poll("on_step_end")
            actor.settle(context, state, hooks)
            if halt:
                state.end_reason = "callback.intermediate_stop"
                break
    finally:
        if opened_actor:
            actor.close(context, state)
        if opened_learner:
            learner.close(context, state)
        hooks.fire("on_episodes_end")
        if top_level:
            hooks.fire("on_end")
    return state


# ---------------------------------------------------------------------------------------------
# learner alone
# ---------------------------------------------------------------------------------------------
def run_learner_only(context: RunContext, learner: LearnerDriver, state: Optional[RunStateTrainer] = None) -> RunStateTrainer:
    context.check_context_parameter()
    context.setup_device()
    if state is None:
        state = RunStateTrainer()
    hooks = HookTable(context.callbacks, context=context, state=state)
    top_level = not context.distributed
    learner.attach(context, state)
    if top_level:
        hooks.fire("on_start")
    learner.open(context, state)
    hooks.fire("on_trainer_start")
    try:
        rules = StopRules(context, True, None)
        state.elapsed_t0 = time.time()
        while True:
            why = rules.reason(state)
            if why:
                state.end_reason = why
                break
            hooks.fire("on_train_before")
            done = learner.update(1, state)
            state.is_step_trained = done > 0
            state.train_count += done
            if hooks.poll("on_train_after"):
                state.end_reason = "callback.trainer_intermediate_stop"
                break
    finally:
        learner.close(context, state)
        hooks.fire("on_trainer_end")
        if top_level:
            hooks.fire("on_end")
    return state


# ---------------------------------------------------------------------------------------------
# plugin drivers: one host environment, the registered RLWorker / RLTrainer classes
# ---------------------------------------------------------------------------------------------
class PluginActor(ActorDriver):
    """One `EnvRun` played by one `WorkerRun` per seat.  Draws from Python's `random` in the reference's order
    (seed draw, per-episode seat shuffle), so that a seeded run interleaves with the plugins' own draws
    identically (tests/test_plugin_surface.py::test_seed_determinism, rollout_items fixtures)."""

    lanes = 1

    def __init__(self, env, worker, workers: Optional[List] = None):
        self.env, self.worker, self.workers = env, worker, workers

    def attach(self, context, state):
        env, plugin = self.env, self.worker.worker
        state.env, state.worker = env, self.worker
        state.parameter, state.memory = plugin.parameter, plugin.memory
        if self.workers is None:
            self.workers, _ = context.rl_config.make_workers(context.players, env, state.parameter, state.memory, self.worker)
        state.workers = self.workers
        assert env.player_num == len(self.workers)

    def open(self, context, state):
        env = self.env
        if context.distributed:
            self.worker.config.setup_from_actor(context.actor_num, context.actor_id)
        if context.seed is not None:
            common.set_seed(context.seed, context.seed_enable_gpu)
            state.episode_seed = random.randint(0, 2 ** (16 - 4))  # core_play.py:79: the first draw of a seeded run
        env.setup(context)
        for w in self.workers:
            w.setup(context, run_state=state)
        state.worker_indices = list(range(env.player_num))
        self._frameskip = lambda: self.workers[state.worker_idx].config.frameskip

    def _seat(self, state, player: int) -> int:
        return state.worker_indices[player]

    def roll_episodes(self, context, state, hooks) -> bool:
        env = self.env
        if not env.done:
            return True
        state.episode_count += 1
        if context.max_episodes > 0 and state.episode_count >= context.max_episodes:
            return False
        env.reset(seed=state.episode_seed)
        if state.episode_seed is not None:
            state.episode_seed += 1
        if context.shuffle_player:
            random.shuffle(state.worker_indices)
        state.worker_idx = self._seat(state, env.next_player)
        for seat, w in enumerate(self.workers):
            w.reset(state.worker_indices[seat], seed=state.episode_seed)
        hooks.fire("on_episode_begin")
        return True

    def act(self, context, state, hooks):
        env = self.env
        state.action = self.workers[state.worker_idx].policy()
        hooks.fire("on_step_action_after")
        if env.done:  # a worker may end the episode from inside policy()
            return
        env.step(state.action, self._frameskip())
        for w in self.workers:
            w.on_step()
        state.total_step += 1

    def _book_episode(self, state):
        env = self.env
        totals = [env.episode_rewards[self._seat(state, p)] for p in range(env.player_num)]
        state.episode_rewards_list.append(totals)
        state.last_episode_rewards = totals
        state.last_episode_step = env.step_num
        state.last_episode_time = env.elapsed_time

    def settle(self, context, state, hooks):
        state.worker_idx = self._seat(state, self.env.next_player)
        if self.env.done:
            self._book_episode(state)
            hooks.fire("on_episode_end")

    def close(self, context, state):
        self.env.teardown()
        for w in self.workers or []:
            w.teardown()
        # a run that stopped inside its first episode still reports that episode's partial return (core_play.py:221-228)
        if state.episode_count == 0 and self.env.step_num > 0 and not state.episode_rewards_list:
            self._book_episode(state)


class PluginLearner(LearnerDriver):
    def __init__(self, trainer):
        self.trainer = trainer

    def attach(self, context, state):
        t = self.trainer
        state.trainer = t
        if getattr(state, "memory", None) is None:
            state.memory = t.memory
        if getattr(state, "parameter", None) is None:
            state.parameter = t.parameter

    def open(self, context, state):
        self.trainer.setup(context)

    def update(self, count: int, state) -> int:
        t = self.trainer
        start = t.train_count
        for _ in range(count):
            t.train()
        return t.train_count - start

    def close(self, context, state):
        self.trainer.teardown()


def play(context: RunContext, env, worker, trainer=None, workers: Optional[List] = None, state: Optional[RunStateActor] = None) -> RunStateActor:
    """The reference's entry point (`core_play.play(context, env, worker, trainer)`, core_play.py:15) on the plugin drivers."""
    learner = None
    if not context.disable_trainer:
        if trainer is None and context.training:
            plugin = worker.worker
            trainer = context.rl_config.make_trainer(plugin.parameter, plugin.memory)
        if trainer is not None:
            learner = PluginLearner(trainer)
    return run_sequence(context, PluginActor(env, worker, workers), learner, state)


def play_trainer_only(context: RunContext, trainer, state: Optional[RunStateTrainer] = None) -> RunStateTrainer:
    """`core_train_only.play_trainer_only(context, trainer)` (core_train_only.py:12) on the plugin learner."""
    return run_learner_only(context, PluginLearner(trainer), state)
