"""Learning-rate / momentum schedules of the shipped configs, evaluated per iteration and pushed into the training step's
device-side optimizer state (TrainStep.set_hyper): a hipGraph-captured step follows them without re-capture.

ref: projects/configs/uni3detr/uni3detr_sunrgbd.py:234-239 (`step` policy, steps at epochs 32 and 38, no warm-up),
     projects/configs/uni3detr/uni3detr_nuscenes.py:301-317 (`cyclic` lr + `cyclic` momentum; upstream mmcv StepLrUpdaterHook /
     CyclicLrUpdaterHook / CyclicMomentumUpdaterHook, restated from their definitions: step -> base * gamma^(#milestones passed),
     linear/constant/exp warm-up; cyclic -> two cosine-annealed phases per cycle, ratio 1 -> target_ratio[0] over the first
     step_ratio_up of the cycle, then -> target_ratio[1]).
"""
import math


def _anneal_cos(start, end, t):
    return end + 0.5 * (start - end) * (math.cos(math.pi * t) + 1.0)


class StepSchedule:
    """policy='step': by_epoch milestones; optional mmcv warm-up ('constant' | 'linear' | 'exp') over the first warmup_iters iterations."""

    def __init__(self, base_lr, step, gamma=0.1, iters_per_epoch=1, warmup=None, warmup_iters=0, warmup_ratio=0.1, min_lr=None):
        self.base_lr, self.gamma, self.ipe = float(base_lr), float(gamma), int(iters_per_epoch)
        self.steps = [step] if isinstance(step, int) else list(step)
        self.warmup, self.warmup_iters, self.warmup_ratio, self.min_lr = warmup, int(warmup_iters), float(warmup_ratio), min_lr

    def lr(self, it):
        epoch = it // self.ipe
        n = sum(1 for s in self.steps if epoch >= s)
        lr = self.base_lr * self.gamma ** n
        if self.min_lr is not None:
            lr = max(lr, self.min_lr)
        if self.warmup is not None and it < self.warmup_iters:
            if self.warmup == "constant":
                lr = lr * self.warmup_ratio
            elif self.warmup == "linear":
                lr = lr * (1 - (1 - it / self.warmup_iters) * (1 - self.warmup_ratio))
            elif self.warmup == "exp":
                lr = lr * self.warmup_ratio ** (1 - it / self.warmup_iters)
            else:
                raise ValueError(self.warmup)
        return lr

    def hyper(self, it):
        return dict(lr=self.lr(it))


class CyclicSchedule:
    """policy='cyclic' for the learning rate and, when momentum_ratio is given, for beta1 (AdamW's `betas[0]`, what mmcv's
    CyclicMomentumUpdaterHook drives)."""

    def __init__(self, base_lr, max_iters, target_ratio=(10, 1e-4), cyclic_times=1, step_ratio_up=0.4, base_betas=(0.9, 0.999),
                 momentum_ratio=None):
        self.base_lr, self.max_iters = float(base_lr), int(max_iters)
        self.base_betas = tuple(base_betas)
        self.cycle = self.max_iters // int(cyclic_times)
        self.up = int(step_ratio_up * self.cycle)
        self.lr_ratio = tuple(target_ratio)
        self.mom_ratio = None if momentum_ratio is None else tuple(momentum_ratio)

    def _ratio(self, it, ratios):
        c = it % self.cycle if self.cycle else 0
        if c < self.up:
            return _anneal_cos(1.0, ratios[0], c / max(1, self.up))
        return _anneal_cos(ratios[0], ratios[1], (c - self.up) / max(1, self.cycle - self.up))

    def lr(self, it):
        return self.base_lr * self._ratio(it, self.lr_ratio)

    def hyper(self, it):
        out = dict(lr=self.lr(it))
        if self.mom_ratio is not None:
            out["betas"] = (self.base_betas[0] * self._ratio(it, self.mom_ratio), self.base_betas[1])
        return out


def build_schedule(cfg, iters_per_epoch):
    """From a loaded config (uni3detr_amd.registry.Config): optimizer.lr / betas, lr_config, momentum_config, runner.max_epochs."""
    opt = cfg["optimizer"]
    lrc = dict(cfg.get("lr_config") or dict(policy="step", step=[]))
    policy = lrc.pop("policy")
    max_epochs = int((cfg.get("runner") or {}).get("max_epochs", cfg.get("total_epochs", 1)))
    betas = tuple(opt.get("betas", (0.9, 0.999)))
    if policy == "step":
        return StepSchedule(opt["lr"], lrc.get("step", []), lrc.get("gamma", 0.1), iters_per_epoch, lrc.get("warmup"),
                            lrc.get("warmup_iters", 0), lrc.get("warmup_ratio", 0.1), lrc.get("min_lr"))
    if policy == "cyclic":
        mom = cfg.get("momentum_config")
        return CyclicSchedule(opt["lr"], max_epochs * iters_per_epoch, lrc.get("target_ratio", (10, 1e-4)), lrc.get("cyclic_times", 1),
                              lrc.get("step_ratio_up", 0.4), betas, None if not mom else mom.get("target_ratio", (0.85 / 0.95, 1)))
    raise NotImplementedError(f"lr policy {policy!r} is not used by any shipped Uni3DETR config")


def apply(ts, schedule, it):
    """Set the hyper-parameters of iteration `it` on a TrainStep (one tiny launch; replayed graphs read them from device memory)."""
    ts.set_hyper(**schedule.hyper(it))
