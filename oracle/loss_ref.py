"""Oracle: train-step composition ``loss_fn`` (R7) and LR schedule.  TEST INFRASTRUCTURE.

``/root/reference/util.py:186-251`` is a syntax-error sketch; this restates its
intent per repair R7 from the reference's own importable stages.  The composition is
"parity unpinned by the reference"; each stage is pinned separately.
``lr_schedule`` restates ``util.py:81-156`` (LinearWarmupCosineDecay).
"""
import math

import torch

from . import features_ref as fr
from . import stft_loss_ref as sl


def loss_fn(net, clean_B1L, noisy_B1L, ell_p_lambda=1.0, stft_lambda=1.0, stft_config=None, pcen=False,
            beta=0.5):
    """features(noisy) -> net -> PHM -> iSTFT -> L1(audio, clean) + stft_lambda*(sc+mag).

    Returns (loss, dict(l1, stft_sc, stft_mag), denoised_audio)."""
    stft_config = dict(stft_config or {})
    stft_config.pop("band", None)
    B, _, L = noisy_B1L.shape
    feats = fr.features_batch(noisy_B1L, pcen=pcen)          # (B*T, C, F)
    T = feats.shape[0] // B
    out = net(feats)                                          # (B*T, 8, F)
    den = fr.denoise_from_output(out, T, beta=beta, length=L)  # (B, L)
    clean = clean_B1L[:, 0]
    l1 = torch.abs(torch.nn.functional.l1_loss(den, clean))   # util.py:239-240
    loss = l1                                                 # util.py:242: unscaled (ell_p_lambda is never used)
    info = {"l1": l1.detach()}
    if stft_lambda > 0:
        sc, mag = sl.mr_stft_loss(den, clean, **stft_config)
        loss = loss + (sc + mag) * stft_lambda
        info["stft_sc"] = sc.detach() * stft_lambda
        info["stft_mag"] = mag.detach() * stft_lambda
    return loss, info, den


def lr_schedule(step, lr_max, n_iter, divider=25, warmup_proportion=0.3):
    """LR returned by the ``step``-th call (1-based) of LinearWarmupCosineDecay.step()
    started at iteration 0 (util.py:110-156), before the wrap-around at n_iter."""
    p1 = int(n_iter * warmup_proportion)
    p2 = n_iter - p1
    lr_min = lr_max / divider
    if step <= p1:
        return lr_min + (step / p1) * (lr_max - lr_min)
    prop = (step - p1) / p2
    end = lr_min / 1e4
    return end + (lr_max - end) / 2 * (math.cos(math.pi * prop) + 1)
