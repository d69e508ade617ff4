"""Per-step low-precision copies ("shadows") of the fp32 master parameters.

In bf16 mode every conv / linear used to cast its own weight (and bias) on every forward: ~170 cast launches per step of a few
microseconds each.  A ShadowSet keeps one persistent bf16 tensor per parameter and refreshes ALL of them with one
multi-tensor copy at the top of the training step; `compute_copy()` hands the shadow out while the set is active and falls
back to an on-the-spot cast otherwise (eval, modules called on their own, parameters modified since the refresh).
"""
import contextlib

import torch

_ACTIVE = [False]


class ShadowSet:
    def __init__(self, params, dtype):
        self.dtype = dtype
        self.params = [p for p in params if p.is_floating_point() and p.dtype != dtype]
        self.shadows = [torch.empty_like(p, dtype=dtype) for p in self.params]
        for p, s in zip(self.params, self.shadows):
            p._u3d_shadow = [s, -1]

    def refresh(self):
        with torch.no_grad():
            torch._foreach_copy_(self.shadows, self.params)
        for p in self.params:
            p._u3d_shadow[1] = p._version

    @contextlib.contextmanager
    def active(self):
        """Refresh, then let compute_copy() use the shadows for the duration of the block (one training forward)."""
        self.refresh()
        prev = _ACTIVE[0]
        _ACTIVE[0] = True
        try:
            yield self
        finally:
            _ACTIVE[0] = prev


def compute_copy(p, dtype):
    """`p` in the compute dtype: its shadow when fresh (no launch), else a cast."""
    if p is None or p.dtype == dtype:
        return p
    if _ACTIVE[0]:
        sh = getattr(p, "_u3d_shadow", None)
        if sh is not None and sh[0].dtype == dtype and sh[1] == p._version:
            return sh[0]
    return p.detach().to(dtype)
