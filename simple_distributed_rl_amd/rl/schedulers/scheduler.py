"""SchedulerConfig (srl/rl/schedulers/scheduler.py) with the constant / linear / cosine schedules
(srl/rl/schedulers/schedulers/*.py): `cfg.set_linear(1.0, 0.1, 1_000_000).create(v).update(step).to_float()`."""
import math
from dataclasses import dataclass, field
from typing import List, Optional


class Scheduler:
    def update(self, step: int) -> "Scheduler":
        raise NotImplementedError()

    def to_float(self) -> float:
        raise NotImplementedError()


class Constant(Scheduler):
    def __init__(self, rate: float, **_):
        self.rate = rate

    def update(self, step: int) -> Scheduler:
        return self

    def to_float(self) -> float:
        return self.rate


class Linear(Scheduler):
    def __init__(self, start_rate: float, end_rate: float, phase_steps: int, **_):
        self.start_rate, self.end_rate, self.phase_steps = start_rate, end_rate, phase_steps
        self.step_rate = (start_rate - end_rate) / phase_steps
        self.rate = start_rate

    def update(self, step: int) -> Scheduler:
        self.rate = self.end_rate if step >= self.phase_steps else self.start_rate - self.step_rate * step
        return self

    def to_float(self) -> float:
        return self.rate


class Cosine(Scheduler):
    def __init__(self, start_rate: float, end_rate: float, phase_steps: int, **_):
        self.start_rate, self.end_rate, self.phase_steps = start_rate, end_rate, phase_steps
        self.rate = start_rate

    def update(self, step: int) -> Scheduler:
        if step >= self.phase_steps:
            self.rate = self.end_rate
        else:
            c = 0.5 * (1 + math.cos(math.pi * step / self.phase_steps))
            self.rate = self.end_rate + (self.start_rate - self.end_rate) * c
        return self

    def to_float(self) -> float:
        return self.rate


class ListScheduler(Scheduler):
    def __init__(self, params: List[dict]):
        self.items = [(SchedulerConfig._create_scheduler(p), int(p.get("phase_steps", p.get("phase_stepsd", 0)))) for p in params]
        self.rate = self.items[0][0].to_float()

    def update(self, step: int) -> Scheduler:
        base = 0
        for sch, steps in self.items:
            if step < base + steps or sch is self.items[-1][0]:
                self.rate = sch.update(step - base).to_float()
                break
            base += steps
        return self

    def to_float(self) -> float:
        return self.rate


@dataclass
class SchedulerConfig:
    schedulers: List[dict] = field(default_factory=list)
    default_scheduler: bool = False

    def clear(self):
        self.schedulers = []
        return self

    def set(self, rate: float):
        return self.clear().add(rate, 0)

    def add(self, rate: float, phase_steps: int = 0):
        self.schedulers.append({"name": "constant", "phase_steps": phase_steps, "rate": rate})
        return self

    def set_linear(self, start_rate: float, end_rate: float, phase_steps: int):
        return self.clear().add_linear(start_rate, end_rate, phase_steps)

    def add_linear(self, start_rate: float, end_rate: float, phase_steps: int):
        self.schedulers.append(dict(name="linear", start_rate=start_rate, end_rate=end_rate, phase_steps=phase_steps))
        return self

    def set_cosine(self, start_rate: float, end_rate: float, phase_steps: int):
        return self.clear().add_cosine(start_rate, end_rate, phase_steps)

    def add_cosine(self, start_rate: float, end_rate: float, phase_steps: int):
        self.schedulers.append(dict(name="cosine", start_rate=start_rate, end_rate=end_rate, phase_steps=phase_steps))
        return self

    def create(self, val: Optional[float] = None) -> Scheduler:
        if self.default_scheduler and val is not None:
            self.set(val)
        if val is not None and len(self.schedulers) == 0:
            self.set(val)
        assert len(self.schedulers) > 0, "Set at least one Scheduler."
        if len(self.schedulers) == 1:
            return self._create_scheduler(self.schedulers[0])
        return ListScheduler(self.schedulers)

    @staticmethod
    def _create_scheduler(params: dict) -> Scheduler:
        name = params["name"]
        if name == "constant":
            return Constant(**params)
        if name == "linear":
            return Linear(**params)
        if name == "cosine":
            return Cosine(**params)
        raise ValueError(name)

    def is_update_step(self) -> bool:
        return len(self.schedulers) > 0 and self.schedulers[-1]["name"] != "constant"
