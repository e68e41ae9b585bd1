"""MI355X-native ``dataset`` front-end: ``ProcessAudio`` and ``pcenfunc`` with the reference's
signatures (``/root/reference/dataset.py:56-76,130-298``), computed by the HIP kernels of fft.hip.
Only the STFT feature path is here; file I/O / augmentation (dataset.py:79-126,301-412) is outside
the hot path."""
import torch
import torch.nn as nn

from . import _lib as L
from ._lib import check, ptr

N_FFT, HOP, BINS = 512, 128, 257


def _need_gpu(x):
    if not x.is_cuda:
        raise L.TrunetHipError("tinyrecurrentunet_amd.dataset runs on MI355X only (got %s)" % x.device)


def stft_features(audio_BL, pcen=False):
    """(B, L) audio -> (B*T, C, 257) features, C = 4 with PCEN (R2 order) else 3.  One launch per batch
    (the reference loops per utterance, D11)."""
    _need_gpu(audio_BL)
    audio_BL = audio_BL.contiguous().float()
    B, Ln = audio_BL.shape
    T = 1 + Ln // HOP
    C = 4 if pcen else 3
    feat = torch.empty((B * T, C, BINS), device=audio_BL.device, dtype=torch.float32)
    mag = torch.empty((B, T, BINS), device=audio_BL.device, dtype=torch.float32) if pcen else None
    tw = L.twiddles(N_FFT, audio_BL.device)
    check(L.lib().trunet_stft_features(ptr(audio_BL), ptr(feat), ptr(mag), ptr(tw), B, Ln, T, C, L.stream()),
          "stft_features")
    if pcen:
        out = feat.view(-1)[BINS:]          # channel 1 of frame 0
        check(L.lib().trunet_pcen(ptr(mag), out.data_ptr(), B, T, C * BINS, 1e-6, 0.025, 0.98, 2.0, 0.5,
                                  L.stream()), "pcen")
    return feat


def pcenfunc(x, eps=1e-6, s=0.025, alpha=0.98, delta=2, r=0.5, training=False):
    """dataset.py:56-76 on a (B, T, F=257) magnitude tensor (out of place in both modes)."""
    _need_gpu(x)
    x = x.contiguous().float()
    B, T, F = x.shape
    assert F == BINS
    out = torch.empty_like(x)
    check(L.lib().trunet_pcen(ptr(x), ptr(out), B, T, BINS, eps, s, alpha, float(delta), r, L.stream()), "pcen")
    return out


class ProcessAudio(nn.Module):
    """dataset.py:130-298.  ``forward`` (1,1,L) -> (T,3,257); ``backward`` (T,3,257) -> (1,L)."""

    def __init__(self, n_fft=512, hop_length=128, sample_rate=48000, min_level_db=-100):
        super().__init__()
        if n_fft != N_FFT or hop_length != HOP:
            raise L.TrunetHipError("the HIP feature kernels are built for n_fft=512, hop=128 (config/tiny.json)")
        self.n_fft, self.hop_length = n_fft, hop_length
        self.n_mels = n_fft // 2 + 1
        self.sample_rate = self.sr = sample_rate
        self.min_level_db = -100.0      # dataset.py:145 overrides the argument
        self.ref_level_db = 25.0

    # elementwise helpers (dataset.py:207-243), kept for API parity; plain tensor expressions
    def amp_to_db(self, magnitude):
        return 20 * torch.log10(torch.clamp(magnitude, min=1e-7)) - self.ref_level_db

    def db_to_amp(self, db_spec):
        return torch.pow(10, db_spec / 20.0)

    def perm(self, tensor):
        return tensor.permute(2, 0, 1)

    def de_perm(self, tensor):
        return tensor.permute(1, 2, 0)

    def norm(self, db_spec):
        return torch.clamp((((db_spec - self.min_level_db) / -self.min_level_db) * 2.) - 1., -1, 1)

    def de_norm(self, norm_spec):
        return (((torch.clamp(norm_spec, -1, 1) + 1.) / 2.) * -self.min_level_db) + self.min_level_db + self.ref_level_db

    def forward(self, audio):
        if audio.dim() != 3 or audio.shape[0] != 1 or audio.shape[1] != 1:
            raise ValueError("ProcessAudio.forward expects (1, 1, L) like the reference (dataset.py:246-272)")
        return stft_features(audio[0])

    def backward(self, denoised_features):
        """dataset.py:275-298: (T, 3, 257) (mag, sin, cos) -> (1, L) by rect-window iSTFT."""
        _need_gpu(denoised_features)
        f = denoised_features.contiguous().float()
        T = f.shape[0]
        # the mask/iSTFT kernel consumes the 8-channel net-output layout; present (mag, sin, cos) as a set whose
        # "noise" angle equals the "mixture" angle => sigmoid(0) = 1/2, compensated by doubling afterwards
        o = torch.zeros((T, 8, BINS), device=f.device, dtype=torch.float32)
        o[:, 0], o[:, 2], o[:, 3] = f[:, 0], f[:, 1], f[:, 2]
        o[:, 6], o[:, 7] = f[:, 1], f[:, 2]
        Ln = (T - 1) * HOP
        frames = torch.empty((1, T, N_FFT), device=f.device, dtype=torch.float32)
        audio = torch.empty((1, Ln), device=f.device, dtype=torch.float32)
        check(L.lib().trunet_mask_istft_fwd(ptr(o), ptr(frames), ptr(audio), None, None,
                                            ptr(L.twiddles(N_FFT, f.device)), 1, T, Ln, 0.5, L.stream()), "mask_istft")
        return audio * 2.0
