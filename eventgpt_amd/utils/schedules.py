"""Learning-rate schedules as pure functions of the step / epoch index.

* ``LinearWarmupCosineLR`` and ``LinearWarmupStepLR`` follow the schedulers that
  ship (unused) with the reference model code (model/common/optim.py:3-62):
  linear warm-up from ``warmup_start_lr`` over ``warmup_steps`` iterations of the
  first epoch, then per-epoch cosine / step decay.
* ``cosine_annealing`` is the per-epoch schedule of the adapter trainers
  (``CosineAnnealingLR(T_max=epochs, eta_min=1e-6)``,
  pipeline/adapter_train/train_hidden_adapter.py:300-304).

Schedulers act on anything exposing ``param_groups`` (torch optimizers and
``parallel.dp.ShardedAdamW``).
"""
from __future__ import annotations

import math


def cosine_annealing(epoch: int, t_max: int, base_lr: float, eta_min: float = 0.0) -> float:
    return eta_min + (base_lr - eta_min) * 0.5 * (1.0 + math.cos(math.pi * epoch / max(t_max, 1)))


def _set_lr(optimizer, lr: float) -> None:
    for g in optimizer.param_groups:
        g["lr"] = lr


class CosineAnnealing:
    """Epoch-stepped cosine annealing with checkpointable state."""

    def __init__(self, optimizer, t_max: int, eta_min: float = 1e-6):
        self.optimizer, self.t_max, self.eta_min = optimizer, t_max, eta_min
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.last_epoch = 0

    def step(self) -> float:
        self.last_epoch += 1
        lr = cosine_annealing(self.last_epoch, self.t_max, self.base_lr, self.eta_min)
        _set_lr(self.optimizer, lr)
        return lr

    def state_dict(self):
        return {"last_epoch": self.last_epoch, "base_lr": self.base_lr}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lr = sd["last_epoch"], sd["base_lr"]
        _set_lr(self.optimizer, cosine_annealing(self.last_epoch, self.t_max, self.base_lr,
                                                 self.eta_min))


class LinearWarmupCosineLR:
    def __init__(self, optimizer, max_epoch: int, min_lr: float, init_lr: float,
                 warmup_steps: int = 0, warmup_start_lr: float = -1):
        self.optimizer, self.max_epoch, self.min_lr, self.init_lr = optimizer, max_epoch, min_lr, init_lr
        self.warmup_steps = warmup_steps
        self.warmup_start_lr = warmup_start_lr if warmup_start_lr >= 0 else init_lr

    def step(self, cur_epoch: int, cur_step: int) -> float:
        if cur_epoch == 0 and cur_step < self.warmup_steps:
            lr = min(self.init_lr, self.warmup_start_lr + (self.init_lr - self.warmup_start_lr)
                     * cur_step / max(self.warmup_steps, 1))
        else:
            lr = (self.init_lr - self.min_lr) * 0.5 * (1.0 + math.cos(
                math.pi * cur_epoch / self.max_epoch)) + self.min_lr
        _set_lr(self.optimizer, lr)
        return lr


class LinearWarmupStepLR:
    def __init__(self, optimizer, max_epoch: int, min_lr: float, init_lr: float,
                 decay_rate: float = 1.0, warmup_start_lr: float = -1, warmup_steps: int = 0):
        self.optimizer, self.max_epoch, self.min_lr, self.init_lr = optimizer, max_epoch, min_lr, init_lr
        self.decay_rate, self.warmup_steps = decay_rate, warmup_steps
        self.warmup_start_lr = warmup_start_lr if warmup_start_lr >= 0 else init_lr

    def step(self, cur_epoch: int, cur_step: int) -> float:
        if cur_epoch == 0 and cur_step < self.warmup_steps:
            lr = min(self.init_lr, self.warmup_start_lr + (self.init_lr - self.warmup_start_lr)
                     * cur_step / max(self.warmup_steps, 1))
        else:
            lr = max(self.init_lr * (self.decay_rate ** cur_epoch), self.min_lr)
        _set_lr(self.optimizer, lr)
        return lr
