"""RunCallback hook names (srl/base/run/callback.py:11-78).  Hooks that cost time when unused
(on_episode_begin/end, on_step_*, on_train_*) are looked up with hasattr by the loops, as in the reference."""


class RunCallback:
    def on_start(self, context, **kwargs) -> None:
        pass

    def on_end(self, context, **kwargs) -> None:
        pass

    def on_episodes_begin(self, context, state, **kwargs) -> None:
        pass

    def on_episodes_end(self, context, state, **kwargs) -> None:
        pass

    def on_trainer_start(self, context, state, **kwargs) -> None:
        pass

    def on_trainer_end(self, context, state, **kwargs) -> None:
        pass

    def on_memory_start(self, context, info, **kwargs) -> None:
        pass

    def on_memory_end(self, context, info, **kwargs) -> None:
        pass
