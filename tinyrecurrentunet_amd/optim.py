"""Fused AdamW + global gradient norm over flat buffers (``torch.optim.AdamW`` + ``clip_grad_norm_(.., 1e9)``
of ``/root/reference/train.py:68,138-140``): one HIP launch for the update instead of ~100 tensor-wise ones."""
import torch

from . import _lib as L
from ._lib import check, ptr


class FusedAdamW:
    """AdamW (torch defaults: betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2) over the parameters that have
    gradients.  ``param_groups[0]["lr"]`` is honoured so ``LinearWarmupCosineDecay`` drives it unchanged."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        self.param_groups = [{"lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay,
                              "params": self.params}]
        self.step_count = 0
        self._flat = None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            p.grad = None

    def _setup(self, active):
        dev = active[0].device
        n = sum(p.numel() for p in active)
        self._key = tuple(id(p) for p in active)
        self._p = torch.empty(n, device=dev, dtype=torch.float32)
        self._g = torch.empty_like(self._p)
        self._m = torch.zeros_like(self._p)
        self._v = torch.zeros_like(self._p)
        self._pv, self._gv = [], []
        off = 0
        for p in active:
            k = p.numel()
            self._pv.append(self._p[off:off + k].view_as(p))
            self._gv.append(self._g[off:off + k].view_as(p))
            off += k
        self._nsq = torch.zeros(1, device=dev, dtype=torch.float32)
        self._flat = True

    @torch.no_grad()
    def step(self):
        active = [p for p in self.params if p.grad is not None]
        if not active:
            return None
        if self._flat is None or self._key != tuple(id(p) for p in active):
            self._setup(active)
        g = self.param_groups[0]
        self.step_count += 1
        torch._foreach_copy_(self._pv, [p.data for p in active])
        torch._foreach_copy_(self._gv, [p.grad for p in active])
        lib = L.lib()
        check(lib.trunet_sumsq(ptr(self._g), self._g.numel(), ptr(self._nsq), L.stream()), "sumsq")
        check(lib.trunet_adamw(ptr(self._p), ptr(self._g), ptr(self._m), ptr(self._v), self._p.numel(), g["lr"],
                               g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self.step_count,
                               L.stream()), "adamw")
        torch._foreach_copy_([p.data for p in active], self._pv)
        return self._nsq   # squared global gradient norm (device tensor; no host sync)
