"""Oracle: multi-resolution STFT loss.  TEST INFRASTRUCTURE.

Restates ``/root/reference/stft_loss.py``: ``stft`` :9-30, ``SpectralConvergenceLoss`` :33-50,
``LogSTFTMagnitudeLoss`` :53-69, ``STFTLoss`` :72-113, ``MultiResolutionSTFTLoss`` :116-166.
"""
import torch
import torch.nn.functional as F


def stft_mag(x_BL, fft_size, hop, win_length, window):
    """stft_loss.py:9-30 -- centred/reflect Hann STFT, sqrt(clamp(re^2+im^2, 1e-7)), (B, frames, bins)."""
    X = torch.stft(x_BL, fft_size, hop, win_length, window, return_complex=True)
    p = X.real ** 2 + X.imag ** 2
    return torch.sqrt(torch.clamp(p, min=1e-7)).transpose(2, 1)


def stft_loss_one(x_BL, y_BL, fft_size, hop, win_length, window):
    """stft_loss.py:91-113 with band == "full": (spectral convergence, log-magnitude L1)."""
    xm = stft_mag(x_BL, fft_size, hop, win_length, window)
    ym = stft_mag(y_BL, fft_size, hop, win_length, window)
    sc = torch.norm(ym - xm, p="fro") / torch.norm(ym, p="fro")
    mag = F.l1_loss(torch.log(ym), torch.log(xm))
    return sc, mag


def mr_stft_loss(x, y, fft_sizes=(512, 1024, 2048), hop_sizes=(50, 120, 240),
                 win_lengths=(240, 600, 1200), sc_lambda=0.5, mag_lambda=0.5):
    """stft_loss.py:141-166.  x = predicted, y = ground truth; (B,L) or (B,C,L)."""
    if x.dim() == 3:
        x = x.reshape(-1, x.size(2))
        y = y.reshape(-1, y.size(2))
    sc_tot, mag_tot = 0.0, 0.0
    for fs, hs, wl in zip(fft_sizes, hop_sizes, win_lengths):
        w = torch.hann_window(wl, dtype=x.dtype, device=x.device)
        sc, mag = stft_loss_one(x, y, fs, hs, wl, w)
        sc_tot = sc_tot + sc
        mag_tot = mag_tot + mag
    n = len(fft_sizes)
    return sc_tot * sc_lambda / n, mag_tot * mag_lambda / n
