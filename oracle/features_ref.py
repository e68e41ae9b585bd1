"""Oracle: STFT feature front-end / inverse, PCEN and phase-aware mask.  TEST INFRASTRUCTURE.

Restates ``/root/reference/dataset.py`` (``ProcessAudio`` :130-298, ``unwrap``/``diff``
:24-51, ``pcenfunc`` :56-76) and ``/root/reference/phm.py`` :7-45 with repair R5.
"""
import math

import torch

N_FFT = 512
HOP = 128
MIN_LEVEL_DB = -100.0   # dataset.py:145
REF_LEVEL_DB = 25.0     # dataset.py:146


def amp_to_db(mag):
    """dataset.py:207-211."""
    return 20.0 * torch.log10(torch.clamp(mag, min=1e-7)) - REF_LEVEL_DB


def db_to_amp(db):
    """dataset.py:214-218."""
    return torch.pow(10.0, db / 20.0)


def norm(db):
    """dataset.py:229-235."""
    return torch.clamp(((db - MIN_LEVEL_DB) / -MIN_LEVEL_DB) * 2.0 - 1.0, -1, 1)


def de_norm(x):
    """dataset.py:238-243."""
    return ((torch.clamp(x, -1, 1) + 1.0) / 2.0) * -MIN_LEVEL_DB + MIN_LEVEL_DB + REF_LEVEL_DB


def stft_rect(audio_1L, n_fft=N_FFT, hop=HOP):
    """dataset.py:260-264: window=None (rectangular), center/reflect, onesided -> (F, T) complex."""
    return torch.stft(audio_1L, n_fft=n_fft, hop_length=hop, normalized=False, return_complex=True)[0]


def features_one(audio_11L, pcen=False, n_fft=N_FFT, hop=HOP):
    """``ProcessAudio.forward`` (dataset.py:246-272) for one utterance (1,1,L) -> (T, C, F).

    ``unwrap`` is the identity on this input, so the phase features are
    sin(angle X), cos(angle X) (dataset.py:173-177; "real" = sin, "imag" = cos).
    With ``pcen`` the 4-channel order of R2 is (norm-dB-mag, PCEN(mag), sin, cos).
    """
    spec = stft_rect(audio_11L[0], n_fft, hop)               # (F, T)
    mag = spec.abs()
    ph = torch.angle(spec)
    chans = [norm(amp_to_db(mag)).t()]
    if pcen:
        chans.append(pcen_ref(mag.t().unsqueeze(0).clone())[0])
    chans += [torch.sin(ph).t(), torch.cos(ph).t()]
    return torch.stack(chans, dim=1)                         # (T, C, F)


def features_batch(audio_B1L, pcen=False):
    """R6: per-utterance features concatenated on the frame axis -> (B*T, C, F)."""
    return torch.cat([features_one(audio_B1L[b:b + 1], pcen) for b in range(audio_B1L.shape[0])], 0)


def pcen_ref(x_BTF, eps=1e-6, s=0.025, alpha=0.98, delta=2.0, r=0.5):
    """``pcenfunc`` (dataset.py:56-76): M[0]=s*x[0]; M[t]=(1-s)M[t-1]+s*x[t];
    (x/(M+eps)^alpha + delta)^r - delta^r.  Out-of-place (the reference's eval branch
    mutates its argument, :75; values are the same)."""
    T = x_BTF.shape[1]
    M = torch.empty_like(x_BTF)
    M[:, 0] = s * x_BTF[:, 0]
    for t in range(1, T):
        M[:, t] = (1 - s) * M[:, t - 1] + s * x_BTF[:, t]
    return (x_BTF / (M + eps).pow(alpha) + delta).pow(r) - delta ** r


def mod_phase(mag_norm_FT, sin_FT, cos_FT):
    """``ProcessAudio.mod_phase`` (dataset.py:182-203) -> complex (F, T)."""
    wrap = torch.arctan2(sin_FT, cos_FT)
    mag = db_to_amp(de_norm(mag_norm_FT))
    return mag * torch.exp(1j * wrap)


def istft_rect(spec_FT, n_fft=N_FFT, hop=HOP, length=None):
    """dataset.py:293-296: rectangular-window iSTFT -> (1, L)."""
    return torch.istft(spec_FT.unsqueeze(0), n_fft=n_fft, hop_length=hop, normalized=False, length=length)


def inverse_features_one(feat_T3F):
    """``ProcessAudio.backward`` (dataset.py:275-298): (T,3,F) -> (1,L)."""
    m, s, c = feat_T3F.permute(1, 2, 0)
    return istft_rect(mod_phase(m, s, c))


def phase_aware_mask(mixture, estimated, beta=0.5):
    """``PhaseAwareMask.forward`` (phm.py:31-45) with R5 (``phase_mix``->``phase_mixture``,
    ``phase_est``->``phase_estimated``): sigmoid(beta*(angle mix - angle est)) * abs(mix)."""
    soft = 1.0 / (1.0 + torch.exp(-beta * (torch.angle(mixture) - torch.angle(estimated))))
    return soft * torch.abs(mixture)


def denoise_from_output(out_N8F, T, beta=0.5, length=None):
    """R7 glue (util.py:217-235 intent): net output (B*T, 8, F) -> denoised audio (B, L).

    Channels 0..3 = "mixture" set, 4..7 = "noise" set (util.py:217).  Per set the
    magnitude is channel 0 and (sin, cos) are the LAST two channels, i.e. the 3-feature
    order of dataset.py:268-270 with one unused slot (channel 1 / 5, the PCEN slot of
    the 4-feature order).  masked = PHM(mix, noise); spec = masked * exp(j angle mix);
    audio = rect-window iSTFT.
    """
    B = out_N8F.shape[0] // T
    auds = []
    for b in range(B):
        o = out_N8F[b * T:(b + 1) * T]                       # (T, 8, F)
        mix = mod_phase(o[:, 0].t(), o[:, 2].t(), o[:, 3].t())
        noi = mod_phase(o[:, 4].t(), o[:, 6].t(), o[:, 7].t())
        masked = phase_aware_mask(mix, noi, beta)
        spec = masked * torch.exp(1j * torch.angle(mix))
        auds.append(istft_rect(spec, length=length))
    return torch.cat(auds, 0)                                # (B, L)
