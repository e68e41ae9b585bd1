"""``phm.PhaseAwareMask`` (``/root/reference/phm.py:7-45`` with repair R5) on the HIP kernels, forward and backward
(the reference's is an ordinary autograd expression of ``torch.abs`` / ``torch.angle`` / ``torch.exp``).  In the train
step the mask is fused with the iSTFT (``util.loss_fn``), which has its own backward; this is the stand-alone module."""
import torch
import torch.nn as nn

from . import _lib as L
from ._lib import check


def _ri(z):
    return torch.view_as_real(z.to(torch.complex64).contiguous())


class _PhaseAwareMaskFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, mixture, estimated, beta):
        m, e = _ri(mixture), _ri(estimated)
        out = torch.empty(mixture.shape, device=mixture.device, dtype=torch.float32)
        check(L.lib().trunet_phm_fwd(m.data_ptr(), e.data_ptr(), out.data_ptr(), out.numel(), float(beta), L.stream()),
              "phm_fwd")
        ctx.save_for_backward(m, e)
        ctx.beta = float(beta)
        ctx.dtypes = (mixture.dtype, estimated.dtype)
        return out

    @staticmethod
    def backward(ctx, g):
        m, e = ctx.saved_tensors
        g = g.contiguous().float()
        gm = torch.empty_like(m) if ctx.needs_input_grad[0] else None
        ge = torch.empty_like(e) if ctx.needs_input_grad[1] else None
        if gm is None and ge is None:
            return None, None, None
        check(L.lib().trunet_phm_bwd(m.data_ptr(), e.data_ptr(), g.data_ptr(), gm.data_ptr() if gm is not None else None,
                                     ge.data_ptr() if ge is not None else None, g.numel(), ctx.beta, L.stream()), "phm_bwd")

        def cx(t, dt):
            if t is None:
                return None
            z = torch.view_as_complex(t)
            return z.to(dt) if dt.is_complex else z.real.to(dt)      # a real input only sees the real part
        return cx(gm, ctx.dtypes[0]), cx(ge, ctx.dtypes[1]), None


class PhaseAwareMask(nn.Module):
    def __init__(self, beta=0.5):
        super().__init__()
        self.beta = beta

    def forward(self, mixture, estimated):
        if not mixture.is_cuda:
            raise L.TrunetHipError("tinyrecurrentunet_amd.phm runs on MI355X only")
        if mixture.shape != estimated.shape:
            raise ValueError("mixture %s and estimated %s must have the same shape" % (tuple(mixture.shape),
                                                                                         tuple(estimated.shape)))
        return _PhaseAwareMaskFn.apply(mixture, estimated, self.beta)
