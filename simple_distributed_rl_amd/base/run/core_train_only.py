"""The learner-only loop (srl/base/run/core_train_only.py:12-98): trainer.train() back to back."""
import time

from simple_distributed_rl_amd.base.context import RunContext, RunStateTrainer


def play_trainer_only(context: RunContext, trainer, state: RunStateTrainer = None):
    context.check_context_parameter()
    context.setup_device()
    callbacks = context.callbacks
    if state is None:
        state = RunStateTrainer()
    state.trainer = trainer
    state.memory = trainer.memory
    state.parameter = trainer.parameter
    trainer.setup(context)
    h_before = [c for c in callbacks if hasattr(c, "on_train_before")]
    h_after = [c for c in callbacks if hasattr(c, "on_train_after")]
    if not context.distributed:
        [c.on_start(context=context, state=state) for c in callbacks]
    [c.on_trainer_start(context=context, state=state) for c in callbacks]
    try:
        state.elapsed_t0 = time.time()
        while True:
            if context.timeout > 0 and (time.time() - state.elapsed_t0) >= context.timeout:
                state.end_reason = "timeout."
                break
            if context.max_train_count > 0 and state.train_count >= context.max_train_count:
                state.end_reason = "max_train_count over."
                break
            [c.on_train_before(context=context, state=state) for c in h_before]
            prev = trainer.train_count
            trainer.train()
            state.is_step_trained = trainer.train_count > prev
            state.train_count = trainer.train_count
            stop = [c.on_train_after(context=context, state=state) for c in h_after]
            if True in stop:
                state.end_reason = "callback.intermediate_stop"
                break
    finally:
        trainer.teardown()
        [c.on_trainer_end(context=context, state=state) for c in callbacks]
        if not context.distributed:
            [c.on_end(context=context, state=state) for c in callbacks]
    return state
