"""CosineAnnealingWarmupRestarts with the reference's schedule
(builder/utils/cosine_annealing_with_warmup_v2.py:9-91): linear warm-up from min_lr to
max_lr*gamma^cycle, cosine back to min_lr, cycle lengths growing by cycle_mult, driven by
an explicit iteration number (``scheduler.step(iteration)``, trainer.py:190)."""
import math

import torch
from torch.optim.lr_scheduler import _LRScheduler


class CosineAnnealingWarmupRestarts(_LRScheduler):
    def __init__(self, optimizer: torch.optim.Optimizer, first_cycle_steps: int, cycle_mult: float = 1.,
                 max_lr: float = 0.1, min_lr: float = 0.001, warmup_steps: int = 0, gamma: float = 1.,
                 last_epoch: int = -1):
        assert warmup_steps < first_cycle_steps
        self.first_cycle_steps, self.cycle_mult = first_cycle_steps, cycle_mult
        self.base_max_lr = self.max_lr = max_lr
        self.min_lr, self.warmup_steps, self.gamma = min_lr, warmup_steps, gamma
        self.cur_cycle_steps, self.cycle, self.step_in_cycle = first_cycle_steps, 0, last_epoch
        super().__init__(optimizer, last_epoch)
        self.base_lrs = []
        for group in self.optimizer.param_groups:      # every group starts at min_lr
            group["lr"] = self.min_lr
            self.base_lrs.append(self.min_lr)

    def _lr_at(self, base):
        s, w = self.step_in_cycle, self.warmup_steps
        if s == -1:
            return base
        if s < w:
            return (self.max_lr - base) * s / w + base
        return base + (self.max_lr - base) * (1 + math.cos(math.pi * (s - w) / (self.cur_cycle_steps - w))) / 2

    def get_lr(self):
        return [self._lr_at(b) for b in self.base_lrs]

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
            self.step_in_cycle += 1
            if self.step_in_cycle >= self.cur_cycle_steps:
                self.cycle += 1
                self.step_in_cycle -= self.cur_cycle_steps
                self.cur_cycle_steps = int((self.cur_cycle_steps - self.warmup_steps) * self.cycle_mult) + self.warmup_steps
        elif epoch >= self.first_cycle_steps:
            if self.cycle_mult == 1.:
                self.step_in_cycle, self.cycle = epoch % self.first_cycle_steps, epoch // self.first_cycle_steps
            else:
                n = int(math.log((epoch / self.first_cycle_steps * (self.cycle_mult - 1) + 1), self.cycle_mult))
                self.cycle = n
                self.step_in_cycle = epoch - int(self.first_cycle_steps * (self.cycle_mult ** n - 1) / (self.cycle_mult - 1))
                self.cur_cycle_steps = self.first_cycle_steps * self.cycle_mult ** n
        else:
            self.cur_cycle_steps, self.step_in_cycle = self.first_cycle_steps, epoch
        self.max_lr = self.base_max_lr * (self.gamma ** self.cycle)
        self.last_epoch = math.floor(epoch)
        for group, lr in zip(self.optimizer.param_groups, self.get_lr()):
            group["lr"] = lr
