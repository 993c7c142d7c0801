"""Callback dispatch for the play loops.

The reference's loops look every optional hook up with `hasattr` once and then iterate bound lists at each
call site (srl/base/run/core_play.py:98-112).  Here one `HookTable` owns that: it resolves each hook name to
the tuple of bound methods of the callbacks that define it (declaration order = the order of
`context.callbacks`, which is the order the reference calls them in) and the loops just `fire` names.
Hook names and keyword arguments are those of srl/base/run/callback.py:11-78."""
from typing import Callable, Dict, Iterable, Tuple


class HookTable:
    __slots__ = ("_callbacks", "_bound", "_kw")

    def __init__(self, callbacks: Iterable, **common_kwargs):
        self._callbacks = tuple(callbacks)
        self._bound: Dict[str, Tuple[Callable, ...]] = {}
        self._kw = common_kwargs

    def bind(self, **common_kwargs) -> "HookTable":
        """Keyword arguments handed to every hook from now on (context=..., state=...)."""
        self._kw = common_kwargs
        return self

    def listeners(self, name: str) -> Tuple[Callable, ...]:
        got = self._bound.get(name)
        if got is None:
            got = tuple(getattr(c, name) for c in self._callbacks if callable(getattr(c, name, None)))
            self._bound[name] = got
        return got

    def wants(self, *names: str) -> bool:
        """True when at least one callback implements one of `names` (lets a loop skip the work that only
        exists to feed a hook, e.g. a device->host read of per-episode results)."""
        return any(self.listeners(n) for n in names)

    def fire(self, name: str) -> None:
        for f in self.listeners(name):
            f(**self._kw)

    def poll(self, name: str) -> bool:
        """Fires `name` on every listener (all of them run, as in the reference) and reports whether any
        returned True -- the "intermediate stop" convention of on_step_end / on_train_after."""
        stop = False
        for f in self.listeners(name):
            if f(**self._kw) is True:
                stop = True
        return stop
