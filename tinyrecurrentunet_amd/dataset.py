"""MI355X-native ``dataset`` module with the reference's names and signatures (``/root/reference/dataset.py``):

* the STFT feature front-end ``ProcessAudio`` / ``pcenfunc`` (:56-76, :130-298) on the HIP kernels of fft.hip;
* the input pipeline: ``DataAugment`` (:79-126), ``CleanNoisyPairDataset`` (:301-390) and
  ``load_CleanNoisyPairDataset`` (:393-412).  The reference augments and mixes inside CPU DataLoader workers through
  torchaudio; here the workers only read and crop (host I/O), the batch goes to the GPU through pinned memory one step
  ahead on a side stream, and gain + both biquads + the clean/noise mix are ONE HIP launch on the whole batch
  (``trunet_augment_mix``, augment.hip).  ``load_CleanNoisyPairDataset`` yields ``(clean, noisy, fileid)`` like the
  reference's loader (train.py:121), already resident in HBM (the ``.cuda()`` of train.py:124-125 is then a no-op).
"""
import math
import os
import random

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import Dataset
from torch.utils.data.distributed import DistributedSampler

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


def unwrap(p, axis=-1):
    """dataset.py:37-51 as it behaves on the tensors the reference passes (>= 3 dims): the phase itself, leading dim
    squeezed (D13: the correction term is identically zero because ``diff`` :24-34 slices only dims 0 and 1)."""
    if p.dim() < 3:
        raise ValueError("unwrap: the reference's diff() only works for >= 3-D tensors (dataset.py:24-34)")
    return p.squeeze(0)


# ----------------------------------------------------------------------------- input pipeline
def _biquad(kind, sr, cutoff, Q=0.7):
    """RBJ cookbook biquad as torchaudio.functional.lowpass_biquad / highpass_biquad build it (the calls of
    dataset.py:124-125), normalised by a0: (b0, b1, b2, a1, a2) in float64."""
    w0 = 2.0 * math.pi * float(cutoff) / float(sr)
    alpha = math.sin(w0) / 2.0 / Q
    c = math.cos(w0)
    b = ((1 - c) / 2, 1 - c, (1 - c) / 2) if kind == "lowpass" else ((1 + c) / 2, -1 - c, (1 + c) / 2)
    a0 = 1 + alpha
    return (b[0] / a0, b[1] / a0, b[2] / a0, -2 * c / a0, (1 - alpha) / a0)


class DataAugment:
    """dataset.py:79-126: gain in [-12, -5) dB, low-pass biquad 7-10 kHz, high-pass biquad 0.8-1.2 kHz (Q = 0.7) on the
    noise signal.  ``draw()`` picks the three parameters with the reference's ``random.choice`` calls in the reference's
    order; ``__call__`` applies them to a (..., L) tensor on the GPU (one launch for all leading rows)."""

    def __init__(self):
        self.min_gain, self.max_gain = -12.0, -5.0
        self.lp_min, self.lp_max = 7000, 10000
        self.hp_min, self.hp_max = 800, 1200
        self.sr = 48000
        self.gains = torch.arange(self.min_gain, self.max_gain, 0.033)
        self.lp_freqs = torch.arange(self.lp_min, self.lp_max, 100)
        self.hp_freqs = torch.arange(self.hp_min, self.hp_max, 50)

    def draw(self):
        """(lp_cutoff, hp_cutoff, gain_db) -- dataset.py:118-120."""
        lp_cutoff = random.choice(self.lp_freqs)
        hp_cutoff = random.choice(self.hp_freqs)
        gain = random.choice(self.gains)
        return float(lp_cutoff), float(hp_cutoff), float(gain)

    def params(self, lp_cutoff, hp_cutoff, gain_db):
        """the 11 numbers trunet_augment_mix takes per signal: linear gain, low-pass and high-pass biquad"""
        return np.array((10.0 ** (gain_db / 20.0),) + _biquad("lowpass", self.sr, lp_cutoff) +
                        _biquad("highpass", self.sr, hp_cutoff), dtype=np.float32)

    def __call__(self, x, params=None):
        _need_gpu(x)
        x = x.contiguous().float()
        rows = x.reshape(-1, x.shape[-1])
        if params is None:
            params = np.stack([self.params(*self.draw())] * rows.shape[0]) if rows.shape[0] else None
        par = torch.as_tensor(params, dtype=torch.float32).reshape(-1, 11).to(x.device).contiguous()
        out = torch.empty_like(rows)
        check(L.lib().trunet_augment_mix(ptr(rows), None, ptr(par), ptr(out), None, rows.shape[0], rows.shape[1],
                                         L.stream()), "augment_mix")
        return out.reshape(x.shape)


def _read_wav(path):
    """mono float32 in [-1, 1) like torchaudio.load(normalize=True) (dataset.py:359-360)"""
    from scipy.io.wavfile import read as wavread
    sr, data = wavread(path)
    if data.ndim > 1:
        data = data[:, 0]
    if data.dtype == np.int16:
        data = data.astype(np.float32) / 32768.0
    elif data.dtype == np.int32:
        data = data.astype(np.float32) / 2147483648.0
    elif data.dtype == np.uint8:
        data = (data.astype(np.float32) - 128.0) / 128.0
    return torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)), sr


class CleanNoisyPairDataset(Dataset):
    """dataset.py:301-390.  ``root/clean/fileid_{i}.wav`` and noise files in ``root/keyboard`` (training; each noise file
    must be exactly crop length, D19), or the DNS no-reverb test pairs (testing).  An element is
    ``(clean (1, L), noise (1, L), fileid, aug_params (11,))``: reading and the random crop happen here (host I/O in the
    DataLoader workers), the augmentation of the noise and ``noisy = clean + noise`` (:368, :380) are applied to the
    whole batch on the GPU by the loader of ``load_CleanNoisyPairDataset``.  ``root = "synthetic:<items>"`` generates
    DNS-shaped pairs instead of reading files (benchmarks, smoke runs; no dataset ships with the reference)."""

    def __init__(self, root="./", subset="training", crop_length_sec=0, sample_rate=48000):
        super().__init__()
        assert subset is None or subset in ["training", "testing"]
        self.root, self.subset = root, subset
        self.aug = DataAugment()
        self.crop_length_sec = crop_length_sec
        self.sample_rate = sample_rate
        self.synthetic = isinstance(root, str) and root.startswith("synthetic:")
        if self.synthetic:
            self.files = ["synthetic_%d" % i for i in range(int(root.split(":", 1)[1]))]
            self.noise_files = list(self.files)
        elif subset == "training":
            n_clean = len(os.listdir(os.path.join(root, "clean")))
            self.files = [os.path.join(root, "clean", "fileid_{}.wav".format(i)) for i in range(n_clean)]
            self.noise_files = sorted(os.listdir(os.path.join(root, "keyboard")))
        elif subset == "testing":
            sortkey = lambda name: "_".join(name.split("_")[-2:])          # DNS test-sample names
            base = os.path.join(root, "datasets/test_set/synthetic/no_reverb")
            clean_files = sorted(os.listdir(os.path.join(base, "clean")), key=sortkey)
            noisy_files = sorted(os.listdir(os.path.join(base, "noisy")), key=sortkey)
            self.files = []
            for c, n in zip(clean_files, noisy_files):
                assert sortkey(c) == sortkey(n)
                self.files.append((os.path.join(base, "clean", c), os.path.join(base, "noisy", n)))
            self.crop_length_sec = 0
        else:
            raise NotImplementedError

    def _synthetic(self, n, length):
        g = np.random.default_rng(n)
        c = 0.1 * g.standard_normal(length + 1)
        clean = (0.5 * (c[1:] + c[:-1])).astype(np.float32)
        return torch.from_numpy(clean), torch.from_numpy((0.05 * g.standard_normal(length)).astype(np.float32))

    def __getitem__(self, n):
        fileid = self.files[n]
        crop_length = int(self.crop_length_sec * self.sample_rate)
        if self.subset == "testing" and not self.synthetic:
            clean, _ = _read_wav(fileid[0])
            noisy, _ = _read_wav(fileid[1])
            assert len(clean) == len(noisy)
            # no augmentation: unit gain and identity "filters" are not expressible as biquads, so the loader mixes
            # nothing for testing items (params = None): the second element already IS the noisy signal
            return clean.unsqueeze(0), noisy.unsqueeze(0), fileid, torch.zeros(11)
        if self.synthetic:
            clean, noise = self._synthetic(n, max(crop_length, 1) + self.sample_rate // 4)
            noise = noise[:crop_length] if crop_length > 0 else noise
            random.choice(self.noise_files)                 # same RNG consumption as the file-based path
        else:
            noise_file = random.choice(self.noise_files)
            clean, sr = _read_wav(fileid)
            noise, _ = _read_wav(os.path.join(self.root, "keyboard", noise_file))
            self.sample_rate = sr
            crop_length = int(self.crop_length_sec * sr)
        params = torch.from_numpy(self.aug.params(*self.aug.draw()))
        assert crop_length < len(clean)
        if crop_length > 0:                                   # random crop in the time domain (dataset.py:376-378)
            start = np.random.randint(low=0, high=len(clean) - crop_length + 1)
            clean = clean[start:start + crop_length]
        if len(noise) != len(clean):
            raise ValueError("noise file must be exactly the crop length (%d samples), got %d (dataset.py:380, D19)"
                             % (len(clean), len(noise)))
        return clean.unsqueeze(0), noise.unsqueeze(0), fileid, params

    def __len__(self):
        return len(self.files)


def _collate_pairs(items):
    clean = torch.stack([it[0] for it in items])
    other = torch.stack([it[1] for it in items])
    params = torch.stack([it[3] for it in items])
    return clean, other, [it[2] for it in items], params


class GpuPairLoader:
    """Iterates a DataLoader of (clean, noise, fileid, params) batches and yields ``(clean, noisy, fileid)`` on the GPU:
    batch k+1 is copied host -> HBM (pinned, non-blocking) and augmented + mixed by trunet_augment_mix on a side stream
    while the consumer trains on batch k; an event orders the hand-over."""

    def __init__(self, loader, mix=True):
        self.loader, self.mix = loader, mix
        self.dataset = loader.dataset
        self.sampler = loader.sampler

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch, side):
        if batch is None:
            return None
        if not torch.cuda.is_available():
            raise L.TrunetHipError("the input pipeline augments on the GPU: no MI355X visible")
        clean, other, fileid, params = batch
        dev = torch.device("cuda", torch.cuda.current_device())
        with torch.cuda.stream(side):
            clean_d = clean.to(dev, non_blocking=True).float().contiguous()
            other_d = other.to(dev, non_blocking=True).float().contiguous()
            if self.mix:
                par_d = params.to(dev, non_blocking=True).float().contiguous()
                B, _, Ln = other_d.shape
                noisy = torch.empty_like(other_d)
                check(L.lib().trunet_augment_mix(ptr(other_d), ptr(clean_d), ptr(par_d), ptr(noisy), None, B, Ln,
                                                 side.cuda_stream), "augment_mix")
            else:
                noisy = other_d
            ev = side.record_event()
        return clean_d, noisy, fileid, ev

    def __iter__(self):
        it = iter(self.loader)
        side = torch.cuda.Stream() if torch.cuda.is_available() else None
        nxt = self._stage(next(it, None), side)
        while nxt is not None:
            clean, noisy, fileid, ev = nxt
            nxt = self._stage(next(it, None), side)
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            clean.record_stream(cur)
            noisy.record_stream(cur)
            yield clean, noisy, fileid


def load_CleanNoisyPairDataset(root, subset, crop_length_sec, batch_size, sample_rate, num_gpus=1, num_workers=4):
    """dataset.py:393-412: same arguments (``**trainset_config`` of config/tiny.json + subset, batch_size, num_gpus);
    DistributedSampler when num_gpus > 1, else shuffle.  Returns an iterable of (clean (B,1,L), noisy (B,1,L), fileid)."""
    dataset = CleanNoisyPairDataset(root=root, subset=subset, crop_length_sec=crop_length_sec, sample_rate=sample_rate)
    kwargs = {"batch_size": batch_size, "num_workers": num_workers, "pin_memory": torch.cuda.is_available(),
              "drop_last": False, "collate_fn": _collate_pairs}
    if num_gpus > 1:
        loader = torch.utils.data.DataLoader(dataset, sampler=DistributedSampler(dataset), **kwargs)
    else:
        loader = torch.utils.data.DataLoader(dataset, sampler=None, shuffle=True, **kwargs)
    return GpuPairLoader(loader, mix=(subset != "testing" or dataset.synthetic))


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

    # elementwise helpers (dataset.py:156-243), kept for API parity; plain tensor expressions (the fused kernels
    # behind forward() / backward() do not call them)
    def get_mag_phase(self, spectrogram):
        """dataset.py:156-159: complex spectrogram -> (|S| without its leading dim, angle S)."""
        return torch.abs(spectrogram).squeeze(0), torch.angle(spectrogram)

    def demod_phase(self, phase):
        """dataset.py:162-179: (sin, cos) of the "demodulated" phase.  The reference's ``unwrap`` (:37-51) is the identity
        on the 3-D / 4-D tensors ProcessAudio passes (its ``diff`` slices only the first two dims) and ends
        in ``squeeze(0)``; the naming is the reference's: real = sin, imag = cos."""
        demodulated = unwrap(phase)
        return torch.sin(demodulated), torch.cos(demodulated)

    def mod_phase(self, magnitude, real_demod, imag_demod):
        """dataset.py:182-203: (norm-dB magnitude, sin, cos) -> complex spectrogram with a leading batch dim."""
        wrap = torch.arctan2(real_demod, imag_demod)
        mag = self.db_to_amp(self.de_norm(magnitude))
        return (mag * torch.exp(1j * wrap)).unsqueeze(0)

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
