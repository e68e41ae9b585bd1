"""Fused AdamW + global gradient norm over flat buffers (``torch.optim.AdamW`` + ``clip_grad_norm_(.., 1e9)``
of ``/root/reference/train.py:68,138-140``): one HIP launch for the norm and one for the update instead of ~100
tensor-wise ones, with ``state_dict()`` / ``load_state_dict()`` in ``torch.optim.AdamW``'s format so the checkpoints of
``train.py:155-162`` (``optimizer_state_dict``) load both ways.

Layouts.  The TRU-Net backward (engine._wg_finish) leaves every parameter gradient as a view of ONE flat tensor and
registers it (``_lib.register_flat_grad``); when the gradients handed to ``step()`` are exactly those views, the
optimizer adopts that layout: parameters are re-pointed (``p.data``) to views of a flat parameter buffer with the same
offsets, and the two kernels run on the engine's gradient buffer in place -- no packing, no copies.  Any other set of
gradients is packed into an own flat buffer first (multi-tensor copies).  Moments and per-parameter step counts survive a
change of layout."""
import torch

from . import _lib as L
from ._lib import check, ptr


class FusedAdamW:
    """AdamW (torch defaults: betas (0.9, 0.999), eps 1e-8, weight_decay 1e-2) over the parameters that have
    gradients.  ``param_groups[0]["lr"]`` is honoured so ``LinearWarmupCosineDecay`` drives it unchanged; one param
    group (the reference has one, train.py:68)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        self.params = [p for p in params if p.requires_grad]
        self.param_groups = [{"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay,
                              "params": self.params}]
        self.state = {}            # id(p) -> {"step": int, "m": view, "v": view}
        self._key = None           # (kind, ids, offsets) of the current layout
        self._nsq = None

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    # ------------------------------------------------------------------ layouts
    def _relayout(self, kind, active, offsets, total, adopt):
        """Flat parameter / moment buffers with ``offsets`` (elements) for ``active``; carries existing moments over.
        adopt: re-point p.data into the flat parameter buffer (the engine layout), else the buffer is a staging copy."""
        dev = active[0].device
        self._p = torch.zeros(total, device=dev, dtype=torch.float32)
        self._m = torch.zeros(total, device=dev, dtype=torch.float32)
        self._v = torch.zeros(total, device=dev, dtype=torch.float32)
        self._pv = []
        for p, o in zip(active, offsets):
            n = p.numel()
            pv = self._p[o:o + n].view_as(p)
            mv, vv = self._m[o:o + n].view_as(p), self._v[o:o + n].view_as(p)
            st = self.state.get(id(p))
            if st is not None:
                mv.copy_(st["m"])
                vv.copy_(st["v"])
                st["m"], st["v"] = mv, vv
            else:
                self.state[id(p)] = {"step": 0, "m": mv, "v": vv}
            if adopt:
                pv.copy_(p.data)
                p.data = pv
            self._pv.append(pv)
        self._adopted = adopt
        self._offsets, self._total = list(offsets), total
        self._key = (kind, tuple(id(p) for p in active), tuple(offsets))
        if self._nsq is None:
            self._nsq = torch.zeros(1, device=dev, dtype=torch.float32)

    def _engine_layout(self, active):
        """(flat gradient, offsets, total) when every gradient is the engine's view of one registered flat tensor"""
        info = L.flat_grad_of(active[0].grad)
        if info is None:
            return None
        flat, layout, total = info
        base = flat.data_ptr()
        offs = []
        for p in active:
            o = layout.get(id(p))
            if o is None or p.grad.data_ptr() != base + 4 * o or not p.grad.is_contiguous():
                return None
            offs.append(o)
        if len(layout) != len(active):
            return None
        return flat, offs, total

    # ------------------------------------------------------------------ step
    @torch.no_grad()
    def step(self):
        active = [p for p in self.params if p.grad is not None]
        if not active:
            return None
        g = self.param_groups[0]
        lib = L.lib()
        L.bump_mutation_epoch()            # the update goes through raw pointers: tensor version counters do not move
        eng = self._engine_layout(active) if active[0].is_cuda else None
        if eng is not None:
            gflat, offs, total = eng
            key = ("engine", tuple(id(p) for p in active), tuple(offs))
            if key != self._key:
                self._relayout("engine", active, offs, total, adopt=True)
            elif any(p.data.data_ptr() != pv.data_ptr() for p, pv in zip(active, self._pv)):
                # something re-assigned p.data (e.g. module.to()): take the new values and re-point
                for p, pv in zip(active, self._pv):
                    if p.data.data_ptr() != pv.data_ptr():
                        pv.copy_(p.data)
                        p.data = pv
        else:
            offs, total = [], 0
            for p in active:
                offs.append(total)
                total += p.numel()
            key = ("packed", tuple(id(p) for p in active), tuple(offs))
            if key != self._key:
                if self._key is not None and self._adopted:
                    for pv, pid in zip(self._pv, self._key[1]):      # leave the adopted buffer: give p.data its own memory
                        for p in self.params:
                            if id(p) == pid:
                                p.data = pv.clone()
                self._relayout("packed", active, offs, total, adopt=False)
                self._g = torch.empty(total, device=active[0].device, dtype=torch.float32)
                self._gv = [self._g[o:o + p.numel()].view_as(p) for p, o in zip(active, offs)]
            torch._foreach_copy_(self._pv, [p.data for p in active])
            torch._foreach_copy_(self._gv, [p.grad for p in active])
            gflat = self._g
        if not gflat.is_cuda:
            raise L.TrunetHipError("FusedAdamW runs on MI355X only (got %s parameters)" % gflat.device)
        check(lib.trunet_sumsq(ptr(gflat), total, ptr(self._nsq), L.stream()), "sumsq")
        # one launch per run of parameters (adjacent in the layout) that share a step count: one run in the usual case
        runs, start, prev = [], 0, None
        for i, p in enumerate(active):
            st = self.state[id(p)]
            st["step"] += 1
            if prev is not None and st["step"] != prev:
                runs.append((start, i, prev))
                start = i
            prev = st["step"]
        runs.append((start, len(active), prev))
        for a, b, step in runs:
            lo = offs[a]
            hi = offs[b - 1] + active[b - 1].numel() if b < len(active) else total
            n = hi - lo
            check(lib.trunet_adamw(self._p.data_ptr() + 4 * lo, gflat.data_ptr() + 4 * lo, self._m.data_ptr() + 4 * lo,
                                   self._v.data_ptr() + 4 * lo, n, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                   g["weight_decay"], step, L.stream()), "adamw")
        if not self._adopted:
            torch._foreach_copy_([p.data for p in active], self._pv)
        return self._nsq   # squared global gradient norm (device tensor; no host sync)

    # ------------------------------------------------------------------ checkpoints (torch.optim.AdamW format)
    def state_dict(self):
        """Same structure as ``torch.optim.AdamW.state_dict()``: per-parameter ``step`` (0-dim float tensor), ``exp_avg``,
        ``exp_avg_sq`` keyed by the parameter's index, one param group."""
        ref = torch.optim.AdamW([torch.zeros(1)], lr=self.param_groups[0]["lr"]).state_dict()["param_groups"][0]
        group = dict(ref)
        g = self.param_groups[0]
        group.update(lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"], weight_decay=g["weight_decay"],
                     params=list(range(len(self.params))))
        state = {}
        for i, p in enumerate(self.params):
            st = self.state.get(id(p))
            if st is not None and st["step"] > 0:
                state[i] = {"step": torch.tensor(float(st["step"])), "exp_avg": st["m"].detach().clone(),
                            "exp_avg_sq": st["v"].detach().clone()}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.params):
            raise ValueError("loaded state dict has a different number of parameter groups / parameters")
        g = self.param_groups[0]
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in groups[0]:
                g[k] = tuple(groups[0][k]) if k == "betas" else groups[0][k]
        index = {pid: i for i, pid in enumerate(groups[0]["params"])}
        self.state, self._key = {}, None
        for pid, st in sd["state"].items():
            p = self.params[index[pid]]
            self.state[id(p)] = {"step": int(float(st["step"])),
                                 "m": st["exp_avg"].to(p.device, torch.float32).clone(),
                                 "v": st["exp_avg_sq"].to(p.device, torch.float32).clone()}
