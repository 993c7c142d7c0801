"""LRSchedulerConfig (srl/rl/schedulers/lr_scheduler.py:5-140): constant / staircase ("step") / exponential / cosine /
piecewise learning-rate schedules, applied to a torch optimizer like `apply_torch_scheduler`."""
import math
from dataclasses import dataclass, field
from typing import List


@dataclass
class LRSchedulerConfig:
    schedule_type: str = ""
    decay_steps: int = 100_000
    decay_rate: float = 0.1
    min_lr: float = 1e-6
    warmup_steps: int = 0
    piecewise_boundaries: List[int] = field(default_factory=lambda: [100000, 110000])
    piecewise_values: List[float] = field(default_factory=lambda: [1.0, 0.5, 0.1])

    def set_constant(self):
        self.schedule_type = ""
        return self

    clear = set_constant

    def set_step(self, decay_steps: int = 100_000, decay_rate: float = 0.1):
        self.schedule_type, self.decay_steps, self.decay_rate = "step", decay_steps, decay_rate
        return self

    def set_exp(self, decay_steps: int = 100_000, decay_rate: float = 0.1):
        self.schedule_type, self.decay_steps, self.decay_rate = "exp", decay_steps, decay_rate
        return self

    def set_cosine(self, decay_steps: int = 100_000, min_lr: float = 1e-6):
        self.schedule_type, self.decay_steps, self.min_lr = "cosine", decay_steps, min_lr
        return self

    def set_piecewise(self, piecewise_boundaries: List[int], piecewise_values: List[float]):
        self.schedule_type, self.piecewise_boundaries, self.piecewise_values = "piecewise", piecewise_boundaries, piecewise_values
        return self

    def factor(self, step: int, lr: float) -> float:
        """Multiplier of the base learning rate at optimizer step `step` (the schedules of :96-125)."""
        t = self.schedule_type
        if t == "step":
            return self.decay_rate ** (step // self.decay_steps)
        if t == "exp":
            return self.decay_rate ** (step / self.decay_steps)
        if t == "cosine":
            alpha = self.min_lr / lr
            x = min(step, self.decay_steps) / self.decay_steps
            return (1 - alpha) * 0.5 * (1 + math.cos(math.pi * x)) + alpha
        if t == "piecewise":
            k = sum(1 for b in self.piecewise_boundaries if step > b)
            return self.piecewise_values[k] / lr
        return 1.0

    def apply_torch_scheduler(self, optimizer):
        """Returns a torch LambdaLR following the schedule (or None for a constant rate)."""
        import torch

        if self.schedule_type == "":
            return None
        lr = optimizer.param_groups[0]["lr"]
        if not (lr > 0 and self.decay_steps > 0 and self.decay_rate > 0 and 0 <= self.min_lr < lr):
            raise ValueError("LRSchedulerConfig: bad parameters")
        return torch.optim.lr_scheduler.LambdaLR(optimizer, lambda step: self.factor(step, lr))
