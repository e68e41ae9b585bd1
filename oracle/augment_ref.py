"""Oracle: noise augmentation of the input pipeline.  TEST INFRASTRUCTURE.

Restates ``DataAugment.__call__`` (``/root/reference/dataset.py:116-126``) and the mix of ``:380``:
``F.gain`` -> ``F.lowpass_biquad(x, sr, cutoff, Q=0.7)`` -> ``F.highpass_biquad(x, sr, cutoff, Q=0.7)``,
``noisy = clean + noise``.  The arithmetic lives in **torchaudio** (unpinned in ``requirements.txt``; absent from this
image, so it cannot be run here): "parity unpinned by the reference" for this stage.  What is restated is torchaudio's
published algorithm: RBJ audio-EQ-cookbook biquads (``w0 = 2 pi f / sr``, ``alpha = sin w0 / (2 Q)``;
low-pass b = ((1-cos)/2, 1-cos, (1-cos)/2), high-pass b = ((1+cos)/2, -(1+cos), (1+cos)/2), a = (1+alpha, -2 cos,
1-alpha)), applied by ``lfilter`` (direct form, zero initial state) with ``clamp=True`` ([-1, 1]) after each filter;
``gain`` multiplies by ``10 ** (gain_db / 20)``.  Computed in float64 with scipy's ``lfilter``.
"""
import math

import numpy as np
from scipy.signal import lfilter


def biquad_coeffs(kind, sr, cutoff, Q=0.7):
    w0 = 2.0 * math.pi * float(cutoff) / float(sr)
    alpha = math.sin(w0) / 2.0 / Q
    c = math.cos(w0)
    if kind == "lowpass":
        b = [(1 - c) / 2, 1 - c, (1 - c) / 2]
    elif kind == "highpass":
        b = [(1 + c) / 2, -1 - c, (1 + c) / 2]
    else:
        raise ValueError(kind)
    a = [1 + alpha, -2 * c, 1 - alpha]
    return np.array(b) / a[0], np.array(a) / a[0]


def augment(noise, sr, gain_db, lp_cutoff, hp_cutoff):
    """noise (..., L) float -> augmented noise, float64"""
    x = np.asarray(noise, dtype=np.float64) * 10.0 ** (float(gain_db) / 20.0)
    b, a = biquad_coeffs("lowpass", sr, lp_cutoff)
    x = np.clip(lfilter(b, a, x, axis=-1), -1.0, 1.0)
    b, a = biquad_coeffs("highpass", sr, hp_cutoff)
    return np.clip(lfilter(b, a, x, axis=-1), -1.0, 1.0)
