"""``phm.PhaseAwareMask`` (``/root/reference/phm.py:7-45`` with repair R5) on the HIP kernel.
Forward only: in training the mask is fused with the iSTFT (``util.loss_fn``), which has its own backward."""
import torch
import torch.nn as nn

from . import _lib as L
from ._lib import check


class PhaseAwareMask(nn.Module):
    def __init__(self, beta=0.5):
        super().__init__()
        self.beta = beta

    def forward(self, mixture, estimated):
        if not mixture.is_cuda:
            raise L.TrunetHipError("tinyrecurrentunet_amd.phm runs on MI355X only")
        m = torch.view_as_real(mixture.to(torch.complex64).contiguous())
        e = torch.view_as_real(estimated.to(torch.complex64).contiguous())
        out = torch.empty(mixture.shape, device=mixture.device, dtype=torch.float32)
        check(L.lib().trunet_phm_fwd(m.data_ptr(), e.data_ptr(), out.data_ptr(), out.numel(), float(self.beta),
                                     L.stream()), "phm_fwd")
        return out
