"""MI355X-native ``stft_loss`` module: same classes / signatures / buffers as
``/root/reference/stft_loss.py`` (``stft`` :9-30, ``SpectralConvergenceLoss`` :33-50,
``LogSTFTMagnitudeLoss`` :53-69, ``STFTLoss`` :72-113, ``MultiResolutionSTFTLoss`` :116-166), computed by the
batched LDS-FFT kernels of fft.hip: the magnitude spectrograms are never materialised, the forward
produces the three sums per resolution and the backward recomputes the frame FFT."""
import torch
import torch.nn as nn

from . import _lib as L
from ._lib import check, ptr


def _padded_window(window, n):
    wl = window.numel()
    w = torch.zeros(n, device=window.device, dtype=torch.float32)
    left = (n - wl) // 2
    w[left:left + wl] = window.float()
    return w


class _STFTMagFn(torch.autograd.Function):
    """x (B, L) -> magnitudes (B, frames, bins); backward recomputes the frame FFT (trunet_stft_mag_bwd)."""

    @staticmethod
    def forward(ctx, x, win, n, hop, wl):
        x = x.contiguous().float()
        B, Ln = x.shape
        tw = L.twiddles(n, x.device)
        out = torch.empty((B, 1 + Ln // hop, n // 2 + 1), device=x.device, dtype=torch.float32)
        check(L.lib().trunet_stft_mag(ptr(x), None, ptr(win), ptr(tw), ptr(out), None, B, Ln, n, hop, L.stream()),
              "stft_mag")
        ctx.save_for_backward(x, win, tw)
        ctx.n, ctx.hop, ctx.wl = n, hop, wl
        return out

    @staticmethod
    def backward(ctx, g):
        x, win, tw = ctx.saved_tensors
        B, Ln = x.shape
        g = g.contiguous().float()
        frames = torch.empty((B, 1 + Ln // ctx.hop, ctx.wl), device=x.device, dtype=torch.float32)
        gx = torch.empty_like(x)
        check(L.lib().trunet_stft_mag_bwd(ptr(x), ptr(win), ptr(tw), ptr(g), ptr(frames), ptr(gx), B, Ln, ctx.n, ctx.hop,
                                          ctx.wl, L.stream()), "stft_mag_bwd")
        return gx, None, None, None, None


def stft(x, fft_size, hop_size, win_length, window):
    """stft_loss.py:9-30: (B, L) -> magnitude spectrogram (B, frames, fft_size // 2 + 1) = sqrt(clamp(re^2 + im^2, 1e-7)),
    differentiable with respect to x like the reference's (an ordinary autograd function there).  The training losses
    below never materialise magnitudes (fused three-sums kernel); this is the reference's stand-alone helper on the
    same LDS FFT."""
    if not x.is_cuda:
        raise L.TrunetHipError("tinyrecurrentunet_amd.stft_loss runs on MI355X only")
    if x.dim() != 2:
        raise ValueError("stft expects (B, L), got %s" % (tuple(x.shape),))
    win = _padded_window(window.to(x.device), fft_size)
    return _STFTMagFn.apply(x, win, fft_size, hop_size, win_length)


class _STFTLossFn(torch.autograd.Function):
    """(x, y) -> (sc, mag) of one resolution; gradient w.r.t. x only (y is the ground truth)."""

    @staticmethod
    def forward(ctx, x, y, win, n, hop, wl):
        x = x.contiguous().float()
        y = y.contiguous().float()
        B, Ln = x.shape
        nfr = 1 + Ln // hop
        tw = L.twiddles(n, x.device)
        part = torch.empty((B * nfr, 3), device=x.device, dtype=torch.float32)
        lib = L.lib()
        check(lib.trunet_stft_loss_fwd(ptr(x), ptr(y), ptr(win), ptr(tw), ptr(part), B, Ln, n, hop, L.stream()),
              "stft_loss_fwd")
        sums = torch.empty(3, device=x.device, dtype=torch.float32)
        check(lib.trunet_reduce_cols(ptr(part), B * nfr, 3, ptr(sums), L.stream()), "reduce_cols")
        count = float(B * nfr * (n // 2 + 1))
        sc = torch.sqrt(sums[0]) / torch.sqrt(sums[1])
        mag = sums[2] / count
        ctx.save_for_backward(x, y, win, tw, sums)
        ctx.n, ctx.hop, ctx.count, ctx.wl = n, hop, count, wl
        return sc, mag

    @staticmethod
    def backward(ctx, g_sc, g_mag):
        x, y, win, tw, sums = ctx.saved_tensors
        B, Ln = x.shape
        coef = torch.stack([g_sc / (torch.sqrt(sums[0]) * torch.sqrt(sums[1])), g_mag / ctx.count]).float().contiguous()
        # overlap-add as a gather over the window's support: no float atomics
        nfr = 1 + Ln // ctx.hop
        frames = torch.empty((B, nfr, ctx.wl), device=x.device, dtype=torch.float32)
        gx = torch.empty_like(x)
        check(L.lib().trunet_stft_loss_bwd_gather(ptr(x), ptr(y), ptr(win), ptr(tw), ptr(coef), ptr(frames), ptr(gx),
                                                  B, Ln, ctx.n, ctx.hop, ctx.wl, L.stream()), "stft_loss_bwd_gather")
        return gx, None, None, None, None, None


class SpectralConvergenceLoss(nn.Module):
    """stft_loss.py:33-50 (on materialised magnitudes; plain tensor expression kept for API parity)."""

    def forward(self, x_mag, y_mag):
        return torch.norm(y_mag - x_mag, p="fro") / torch.norm(y_mag, p="fro")


class LogSTFTMagnitudeLoss(nn.Module):
    """stft_loss.py:53-69."""

    def forward(self, x_mag, y_mag):
        return torch.nn.functional.l1_loss(torch.log(y_mag), torch.log(x_mag))


class STFTLoss(nn.Module):
    """stft_loss.py:72-113 (band == "full"; "high" is a frame-axis slice in the reference, D17: unsupported)."""

    def __init__(self, fft_size=1024, shift_size=120, win_length=600, window="hann_window", band="full"):
        super().__init__()
        self.fft_size, self.shift_size, self.win_length, self.band = fft_size, shift_size, win_length, band
        self.spectral_convergence_loss = SpectralConvergenceLoss()
        self.log_stft_magnitude_loss = LogSTFTMagnitudeLoss()
        self.register_buffer("window", getattr(torch, window)(win_length))
        self._wpad = None

    def forward(self, x, y):
        if self.band != "full":
            raise NotImplementedError("band=%r (stft_loss.py:103-110)" % self.band)
        if not x.is_cuda:
            raise L.TrunetHipError("tinyrecurrentunet_amd.stft_loss runs on MI355X only")
        if self._wpad is None or self._wpad.device != x.device:
            self._wpad = _padded_window(self.window.to(x.device), self.fft_size)
        return _STFTLossFn.apply(x, y, self._wpad, self.fft_size, self.shift_size, self.win_length)


class MultiResolutionSTFTLoss(nn.Module):
    """stft_loss.py:116-166."""

    def __init__(self, fft_sizes=[1024, 2048, 512], hop_sizes=[120, 240, 50], win_lengths=[600, 1200, 240],
                 window="hann_window", sc_lambda=0.1, mag_lambda=0.1, band="full"):
        super().__init__()
        self.sc_lambda, self.mag_lambda = sc_lambda, mag_lambda
        assert len(fft_sizes) == len(hop_sizes) == len(win_lengths)
        self.stft_losses = nn.ModuleList()
        for fs, ss, wl in zip(fft_sizes, hop_sizes, win_lengths):
            self.stft_losses += [STFTLoss(fs, ss, wl, window, band)]

    def forward(self, x, y):
        if x.dim() == 3:
            x = x.reshape(-1, x.size(2))
            y = y.reshape(-1, y.size(2))
        sc_loss, mag_loss = 0.0, 0.0
        for f in self.stft_losses:
            sc_l, mag_l = f(x, y)
            sc_loss = sc_loss + sc_l
            mag_loss = mag_loss + mag_l
        n = len(self.stft_losses)
        return sc_loss * self.sc_lambda / n, mag_loss * self.mag_lambda / n
