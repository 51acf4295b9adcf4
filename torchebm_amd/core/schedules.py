"""Host-side parameter schedules and the ``Schedulable`` mixin.

Mirror of the reference's torchebm/core/base_scheduler.py and core/schedulable.py.
Schedules are plain Python floats evaluated on the host; they are the only way scalars
enter the sampler loops.  The value used by iteration ``i`` of a
sampler is the one at ``step_count == i`` (read before ``step_schedulers()``).

For the k-fused kernels the schedule is *pre-expanded*: ``preview(k)`` returns the k
values the next k iterations would read, without touching the scheduler, and
``advance(k)`` afterwards leaves it exactly where k ``step()`` calls would have.
"""

from __future__ import annotations

import copy
import math
from abc import ABC, abstractmethod
from typing import Any, Dict, List, Sequence, Union


class BaseScheduler(ABC):
    """A function of the step count, with ``step`` / ``reset`` / ``get_value`` and
    ``state_dict`` round-tripping (base_scheduler.py:71-280)."""

    def __init__(self, start_value: float):
        if not isinstance(start_value, (float, int)):
            raise TypeError(
                f"{type(self).__name__} received an invalid start_value of type "
                f"{type(start_value).__name__}. Expected float or int."
            )
        self.start_value = float(start_value)
        self.current_value = self.start_value
        self.step_count = 0

    @abstractmethod
    def _compute_value(self) -> float:
        """Value at the current ``self.step_count``."""

    def step(self) -> float:
        self.step_count += 1
        self.current_value = self._compute_value()
        return self.current_value

    def reset(self) -> None:
        self.step_count = 0
        self.current_value = self.start_value

    def get_value(self) -> float:
        return self.current_value

    def state_dict(self) -> Dict[str, Any]:
        return dict(self.__dict__)

    def load_state_dict(self, state_dict: Dict[str, Any]) -> None:
        self.__dict__.update(state_dict)

    # ---- pre-expansion for the fused kernels -------------------------------------
    def is_constant(self) -> bool:
        """True when every future value equals the current one."""
        return False

    def preview(self, k: int) -> List[float]:
        """Values iterations 0..k-1 from now would read; the scheduler is not modified."""
        if self.is_constant():
            return [self.current_value] * k
        ghost = copy.deepcopy(self)
        out = []
        for _ in range(k):
            out.append(ghost.get_value())
            ghost.step()
        return out

    def advance(self, k: int) -> None:
        """Equivalent to ``k`` calls of :meth:`step`."""
        for _ in range(k):
            self.step()


class ConstantScheduler(BaseScheduler):
    def _compute_value(self) -> float:
        return self.start_value

    def is_constant(self) -> bool:
        return self.current_value == self.start_value

    def advance(self, k: int) -> None:
        if self.is_constant():
            self.step_count += k
        else:
            super().advance(k)


class ExponentialDecayScheduler(BaseScheduler):
    """``max(min_value, start * decay_rate**t)``."""

    def __init__(self, start_value: float, decay_rate: float, min_value: float = 0.0):
        super().__init__(start_value)
        if not 0.0 < decay_rate <= 1.0:
            raise ValueError(f"decay_rate must be in (0, 1], got {decay_rate}")
        if min_value < 0:
            raise ValueError(f"min_value must be non-negative, got {min_value}")
        self.decay_rate = decay_rate
        self.min_value = min_value

    def _compute_value(self) -> float:
        return max(self.min_value, self.start_value * self.decay_rate**self.step_count)


class LinearScheduler(BaseScheduler):
    """Linear ramp from ``start_value`` to ``end_value`` over ``n_steps``, then flat."""

    def __init__(self, start_value: float, end_value: float, n_steps: int):
        super().__init__(start_value)
        if n_steps <= 0:
            raise ValueError(f"n_steps must be positive, got {n_steps}")
        self.end_value = end_value
        self.n_steps = n_steps
        self.step_size = (end_value - start_value) / n_steps

    def _compute_value(self) -> float:
        if self.step_count >= self.n_steps:
            return self.end_value
        return self.start_value + self.step_size * self.step_count


class CosineScheduler(BaseScheduler):
    """Half-cosine from ``start_value`` to ``end_value`` over ``n_steps``, then flat."""

    def __init__(self, start_value: float, end_value: float, n_steps: int):
        super().__init__(start_value)
        if n_steps <= 0:
            raise ValueError(f"n_steps must be a positive integer, got {n_steps}")
        self.end_value = end_value
        self.n_steps = n_steps

    def _compute_value(self) -> float:
        if self.step_count >= self.n_steps:
            return self.end_value
        frac = self.step_count / self.n_steps
        blend = 0.5 * (1 + math.cos(math.pi * frac))
        return self.end_value + (self.start_value - self.end_value) * blend


class MultiStepScheduler(BaseScheduler):
    """``start * gamma**(number of milestones reached)``."""

    def __init__(self, start_value: float, milestones: Sequence[int], gamma: float = 0.1):
        super().__init__(start_value)
        if any(m <= 0 for m in milestones):
            raise ValueError("Milestone steps must be positive integers.")
        if any(a >= b for a, b in zip(milestones, milestones[1:])):
            raise ValueError("Milestones must be strictly increasing.")
        self.milestones = sorted(milestones)
        self.gamma = gamma

    def _compute_value(self) -> float:
        reached = sum(1 for m in self.milestones if self.step_count >= m)
        return self.start_value * self.gamma**reached


class WarmupScheduler(BaseScheduler):
    """Linear warm-up to ``main_scheduler.start_value`` over ``warmup_steps``, then the
    main scheduler (which only starts stepping once the warm-up is over)."""

    def __init__(self, main_scheduler: BaseScheduler, warmup_steps: int, warmup_init_factor: float = 0.01):
        super().__init__(main_scheduler.start_value * warmup_init_factor)
        self.main_scheduler = main_scheduler
        self.warmup_steps = warmup_steps
        self.warmup_init_factor = warmup_init_factor
        self.target_value = main_scheduler.start_value
        self.main_scheduler.reset()

    def _compute_value(self) -> float:
        if self.step_count <= self.warmup_steps:
            frac = self.step_count / self.warmup_steps
            return self.start_value + frac * (self.target_value - self.start_value)
        return self.main_scheduler.current_value

    def step(self) -> float:
        self.step_count += 1
        if self.step_count > self.warmup_steps:
            self.main_scheduler.step()
        self.current_value = self._compute_value()
        return self.current_value

    def reset(self) -> None:
        super().reset()
        self.main_scheduler.reset()


class TemperatureScheduler(BaseScheduler):
    """Energy-matching temperature sweep: epsilon(t) is 0 below ``tau_star``, ramps
    linearly to ``epsilon_max`` at t = 1; the value is sqrt(epsilon) when ``sqrt``
    (base_scheduler.py:857-969).  ``t`` moves from ``t_start`` to ``t_end`` in ``n_steps``."""

    def __init__(
        self,
        epsilon_max: float,
        tau_star: float = 0.8,
        n_steps: int = 200,
        t_start: float = 0.0,
        t_end: float = 1.0,
        sqrt: bool = True,
    ):
        if epsilon_max < 0:
            raise ValueError(f"epsilon_max must be >= 0, got {epsilon_max}")
        if not 0.0 <= tau_star < 1.0:
            raise ValueError(f"tau_star must be in [0, 1), got {tau_star}")
        if n_steps <= 0:
            raise ValueError(f"n_steps must be positive, got {n_steps}")
        if t_end < t_start:
            raise ValueError(f"t_end ({t_end}) must be >= t_start ({t_start})")
        self.epsilon_max = float(epsilon_max)
        self.tau_star = float(tau_star)
        self.n_steps = int(n_steps)
        self.t_start = float(t_start)
        self.t_end = float(t_end)
        self.sqrt = bool(sqrt)
        super().__init__(self._value_at_time(self.t_start))

    def epsilon_at(self, t: float) -> float:
        if t < self.tau_star:
            return 0.0
        if t < 1.0:
            return self.epsilon_max * (t - self.tau_star) / (1.0 - self.tau_star)
        return self.epsilon_max

    def _value_at_time(self, t: float) -> float:
        eps = self.epsilon_at(t)
        return math.sqrt(eps) if self.sqrt else eps

    def _compute_value(self) -> float:
        frac = min(self.step_count, self.n_steps) / self.n_steps
        return self._value_at_time(self.t_start + (self.t_end - self.t_start) * frac)


class Schedulable:
    """Mixin for ``nn.Module`` hosts: named schedulers, stepped/reset over the whole
    module subtree (schedulable.py:17-75)."""

    def __init__(self, *args: Any, **kwargs: Any):
        super().__init__(*args, **kwargs)
        self.schedulers: Dict[str, BaseScheduler] = {}

    def register_scheduler(self, name: str, scheduler: BaseScheduler) -> None:
        self.schedulers[name] = scheduler

    def _register_param(self, name: str, value: Union[float, BaseScheduler], *, positive: bool = False) -> None:
        if isinstance(value, BaseScheduler):
            self.schedulers[name] = value
            return
        if positive and value <= 0:
            raise ValueError(f"{name} must be positive")
        self.schedulers[name] = ConstantScheduler(float(value))

    def get_schedulers(self) -> Dict[str, BaseScheduler]:
        return self.schedulers

    def get_scheduled_value(self, name: str) -> float:
        try:
            return self.schedulers[name].get_value()
        except KeyError:
            raise KeyError(f"No scheduler registered for parameter '{name}'") from None

    def _subtree_schedulers(self) -> List[BaseScheduler]:
        found: List[BaseScheduler] = []
        for module in self.modules():  # nn.Module API; the host must be an nn.Module
            if isinstance(module, Schedulable):
                found.extend(module.schedulers.values())
        return found

    def step_schedulers(self) -> None:
        for sched in self._subtree_schedulers():
            sched.step()

    def reset_schedulers(self) -> None:
        for sched in self._subtree_schedulers():
            sched.reset()

    def advance_schedulers(self, k: int) -> None:
        """``k`` x :meth:`step_schedulers` (used after a k-fused kernel launch)."""
        for sched in self._subtree_schedulers():
            sched.advance(k)
