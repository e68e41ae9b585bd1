"""MI355X-native ``util``: the train-step composition ``loss_fn`` (``/root/reference/util.py:186-251`` in the
repaired form R7 of SURVEY.md section 0.2), the LR schedule (``util.py:81-156``) and the small host helpers
``train.py`` imports (``find_max_epoch`` :30-49, ``print_size`` :52-61, ``rescale`` :26-27).

loss_fn: features(noisy) [HIP STFT] -> net [HIP body] -> phase-aware mask + iSTFT [HIP] ->
L1(audio, clean) + stft_lambda * (sc + mag) [HIP multi-resolution STFT loss]; every stage has a hand-written
backward, chained through ``torch.autograd.Function`` nodes (autograd is glue only)."""
import os
from math import cos, pi

import numpy as np
import torch

from . import _lib as L
from . import dataset as ds
from ._lib import check, ptr

BINS, N_FFT, HOP = 257, 512, 128


class _MaskISTFTL1Fn(torch.autograd.Function):
    """net output (B*T, 8, 257), clean (B, L) -> denoised audio (B, L), l1 = mean |audio - clean|."""

    @staticmethod
    def forward(ctx, net_out, clean, T, beta):
        net_out = net_out.contiguous()
        B = net_out.shape[0] // T
        Ln = (T - 1) * HOP
        dev = net_out.device
        lib = L.lib()
        frames = torch.empty((B, T, N_FFT), device=dev, dtype=torch.float32)
        audio = torch.empty((B, Ln), device=dev, dtype=torch.float32)
        npart = lib.trunet_mask_istft_l1_nparts(B, Ln)
        part = torch.empty(npart, device=dev, dtype=torch.float32)
        tw = L.twiddles(N_FFT, dev)
        check(lib.trunet_mask_istft_fwd(ptr(net_out), ptr(frames), ptr(audio), ptr(clean), ptr(part), ptr(tw), B, T,
                                        Ln, beta, L.stream()), "mask_istft_fwd")
        s = torch.empty(1, device=dev, dtype=torch.float32)
        check(lib.trunet_reduce_cols(ptr(part), npart, 1, ptr(s), L.stream()), "reduce_cols")
        l1 = s[0] / float(B * Ln)
        ctx.save_for_backward(net_out, clean, audio, tw)
        ctx.T, ctx.beta = T, beta
        return audio, l1

    @staticmethod
    def backward(ctx, g_audio, g_l1):
        net_out, clean, audio, tw = ctx.saved_tensors
        T = ctx.T
        B, Ln = audio.shape
        lib = L.lib()
        if g_l1 is None:
            g_l1 = torch.zeros((), device=audio.device)
        scale = (g_l1 / float(B * Ln)).reshape(1).float().contiguous()
        g = torch.empty_like(audio)
        check(lib.trunet_l1_grad(ptr(audio), ptr(clean), ptr(scale), ptr(g), audio.numel(), L.stream()), "l1_grad")
        if g_audio is not None:
            g = g + g_audio
        g_net = torch.empty_like(net_out)
        check(lib.trunet_mask_istft_bwd(ptr(g.contiguous()), ptr(net_out), ptr(g_net), ptr(tw), B, T, Ln, ctx.beta,
                                        L.stream()), "mask_istft_bwd")
        return g_net, None, None, None


def denoise(net_out, clean_BL, T, beta=0.5):
    """R7: (B*T, 8, 257) -> (audio (B, L), l1)."""
    return _MaskISTFTL1Fn.apply(net_out, clean_BL.contiguous().float(), T, float(beta))


# Train-step fast path (round 4): the whole tail of loss_fn -- mask + iSTFT + L1 + the multi-resolution STFT loss -- as ONE
# autograd node.  Composed from the stand-alone pieces (denoise() + MultiResolutionSTFTLoss) the same arithmetic takes ~75
# launches per step, ~65 of them single-element torch kernels (sqrt, div, mul, add, stack ... of the scalar algebra and of its
# autograd) -- 0.3 ms of a 21 ms bf16 step.  Here: 2 + nres launches forward (frames, overlap-add + L1 partial sums, one
# forward-and-gradient-frames kernel per resolution) + 1 (all column sums and the scalar algebra), and 2 backward (one gather
# for d loss / d audio over all resolutions and the L1 term, the mask + iSTFT backward).  TRUNET_FUSED_LOSS=0 keeps the
# composition (the reference's own structure, util.py:239-250), which the fused path is tested against.
FUSED_LOSS = os.environ.get("TRUNET_FUSED_LOSS", "1") != "0"
_LOSS_SCRATCH = {}


class _FusedLossFn(torch.autograd.Function):
    """net output (B*T, 8, 257), clean (B, L) -> (loss, vals); vals = [loss, l1, stft_sc, stft_mag, ...] (no gradient)."""

    @staticmethod
    def forward(ctx, net_out, clean, T, beta, stft_lambda, res, sc_lambda, mag_lambda):
        # res: [(padded window, n, hop, win_length), ...] (empty: L1 only)
        net_out = net_out.contiguous()
        B = net_out.shape[0] // T
        Ln = (T - 1) * HOP
        dev = net_out.device
        lib, st = L.lib(), L.stream()
        frames = torch.empty((B, T, N_FFT), device=dev, dtype=torch.float32)
        audio = torch.empty((B, Ln), device=dev, dtype=torch.float32)
        npart = lib.trunet_mask_istft_l1_nparts(B, Ln)
        part = torch.empty(npart, device=dev, dtype=torch.float32)
        tw = L.twiddles(N_FFT, dev)
        check(lib.trunet_mask_istft_fwd(ptr(net_out), ptr(frames), ptr(audio), ptr(clean), ptr(part), ptr(tw), B, T,
                                        Ln, beta, st), "mask_istft_fwd")
        a = L.LossArgs()
        a.l1_partials, a.n_l1, a.l1_count, a.nres = ptr(part), npart, float(B * Ln), len(res)
        a.sc_lambda, a.mag_lambda, a.stft_lambda = sc_lambda, mag_lambda, stft_lambda
        need_grad = ctx.needs_input_grad[0]
        keep = [part]
        planes = []
        for i, (win, n, hop, wl) in enumerate(res):
            nfr = 1 + Ln // hop
            rp = torch.empty((B * nfr, 3), device=dev, dtype=torch.float32)
            twn = L.twiddles(n, dev)
            if need_grad:
                fs = torch.empty((B, nfr, wl), device=dev, dtype=torch.float32)
                fm = torch.empty((B, nfr, wl), device=dev, dtype=torch.float32)
                check(lib.trunet_stft_loss_fwdgrad(ptr(audio), ptr(clean), ptr(win), ptr(twn), ptr(rp), ptr(fs), ptr(fm), B, Ln,
                                                   n, hop, wl, st), "stft_loss_fwdgrad")
                planes.append((fs, fm, n, hop, wl))
            else:
                check(lib.trunet_stft_loss_fwd(ptr(audio), ptr(clean), ptr(win), ptr(twn), ptr(rp), B, Ln, n, hop, st),
                      "stft_loss_fwd")
            a.parts[i], a.nrows[i], a.count[i] = ptr(rp), B * nfr, float(B * nfr * (n // 2 + 1))
            keep.append(rp)
        scratch = _LOSS_SCRATCH.get(str(dev))
        if scratch is None:
            scratch = _LOSS_SCRATCH[str(dev)] = torch.zeros(lib.trunet_loss_scratch_bytes() // 8, device=dev, dtype=torch.float64)
        loss = torch.empty((), device=dev, dtype=torch.float32)
        vals = torch.empty(5 + 2 * L.MAX_RES, device=dev, dtype=torch.float32)
        check(lib.trunet_loss_finalize(a, ptr(loss), ptr(vals), scratch.data_ptr(), st), "loss_finalize")
        ctx.save_for_backward(net_out, clean, audio, tw, vals)
        ctx.planes, ctx.T, ctx.beta = planes, T, beta
        ctx.mark_non_differentiable(vals)
        return loss, vals

    @staticmethod
    def backward(ctx, g_loss, _g_vals):
        net_out, clean, audio, tw, vals = ctx.saved_tensors
        B, Ln = audio.shape
        lib, st = L.lib(), L.stream()
        g_loss = g_loss.reshape(1).float().contiguous()
        a = L.LossGatherArgs()
        a.nres = len(ctx.planes)
        for i, (fs, fm, n, hop, wl) in enumerate(ctx.planes):
            a.fr_sc[i], a.fr_mag[i], a.n[i], a.hop[i], a.win_length[i] = ptr(fs), ptr(fm), n, hop, wl
        g = torch.empty_like(audio)
        check(lib.trunet_loss_grad_gather(a, ptr(audio), ptr(clean), ptr(vals), ptr(g_loss), ptr(g), B, Ln, st), "loss_grad_gather")
        g_net = torch.empty_like(net_out)
        check(lib.trunet_mask_istft_bwd(ptr(g), ptr(net_out), ptr(g_net), ptr(tw), B, ctx.T, Ln, ctx.beta, st), "mask_istft_bwd")
        ctx.planes = None
        return g_net, None, None, None, None, None, None, None


def _fused_loss_plan(mrstftloss, stft_lambda, dev):
    """[(padded window, n, hop, win_length)] when the loss can take the fused node: our own MultiResolutionSTFTLoss with
    band "full" (or no STFT term at all), at most TRUNET_MAX_RES resolutions; None: compose from the stand-alone pieces"""
    from . import stft_loss as sl
    if not FUSED_LOSS:
        return None
    if not (stft_lambda > 0):
        return []
    if type(mrstftloss) is not sl.MultiResolutionSTFTLoss or len(mrstftloss.stft_losses) > L.MAX_RES:
        return None
    plan = []
    for f in mrstftloss.stft_losses:
        if type(f) is not sl.STFTLoss or f.band != "full":
            return None
        if f._wpad is None or f._wpad.device != dev:
            f._wpad = sl._padded_window(f.window.to(dev), f.fft_size)
        plan.append((f._wpad, f.fft_size, f.shift_size, f.win_length))
    return plan


def loss_fn(net, X, ell_p, ell_p_lambda, stft_lambda, mrstftloss, pcen=None, **kwargs):
    """util.py:186-251 (R7).  X = (clean_audio, noisy_audio), each (B, 1, L) (a leading batch-1 dim as the
    reference's DataLoader gives, util.py:207, is squeezed).  Returns (loss, {"l1", "stft_sc", "stft_mag"})."""
    clean_audio, noisy_audio = X
    if clean_audio.dim() == 4:
        clean_audio, noisy_audio = clean_audio.squeeze(0), noisy_audio.squeeze(0)
    clean = clean_audio[:, 0].contiguous()
    noisy = noisy_audio[:, 0].contiguous()
    if pcen is None:
        pcen = net.encoder[0].StandardConv1d[0].in_channels == 4
    feats = ds.stft_features(noisy, pcen=pcen)
    T = feats.shape[0] // noisy.shape[0]
    # use_tgru (extension): the time-recurrent block needs to know where utterances begin
    out = net(feats, frames_per_seq=T) if getattr(net, "use_tgru", False) else net(feats)
    plan = _fused_loss_plan(mrstftloss, stft_lambda, out.device) if out.is_cuda else None
    if plan is not None:
        loss, vals = _FusedLossFn.apply(out, clean.float(), T, 0.5, float(stft_lambda) if plan else 0.0, plan,
                                        float(mrstftloss.sc_lambda) if plan else 0.0,
                                        float(mrstftloss.mag_lambda) if plan else 0.0)
        output_dic = {"l1": vals[1]}
        if plan:
            output_dic["stft_sc"], output_dic["stft_mag"] = vals[2], vals[3]
        return loss, output_dic
    den, l1 = denoise(out, clean, T)
    l1 = torch.abs(l1)
    loss = l1                  # util.py:239-242: the L1 term enters unscaled (ell_p / ell_p_lambda are accepted and unused)
    output_dic = {"l1": l1.detach()}
    if stft_lambda > 0:
        sc_loss, mag_loss = mrstftloss(den, clean)
        loss = loss + (sc_loss + mag_loss) * stft_lambda
        output_dic["stft_sc"] = sc_loss.detach() * stft_lambda
        output_dic["stft_mag"] = mag_loss.detach() * stft_lambda
    return loss, output_dic


# ----------------------------------------------------------------------------- LR schedule (util.py:110-156)
class LinearWarmupCosineDecay:
    """Learning-rate schedule of train.py:102-110 as a closed form of ONE counter.

    A cycle has ``n_iter`` steps: the first ``int(n_iter * warmup_proportion)`` rise from lr_max/divider to lr_max, the
    rest fall to lr_max/divider/1e4; ``phase`` names the two curve shapes ("linear" or "cosine" each, as in the reference).
    ``step()`` returns the rate of the next iteration, writes it into every param group and wraps around at the end of
    the cycle; ``iteration`` resumes inside a cycle.  Pinned by tests/golden/sched.npz (values of the reference class)."""

    def __init__(self, optimizer, lr_max, n_iter, iteration=0, divider=25, warmup_proportion=0.3,
                 phase=("linear", "cosine")):
        for name in phase:
            if name not in ("linear", "cosine"):
                raise KeyError(name)
        self.optimizer = optimizer
        self.shapes = tuple(phase)
        self.n_iter = n_iter
        self.n_warm = int(n_iter * warmup_proportion)
        self.lr_max, self.lr_low, self.lr_end = lr_max, lr_max / divider, lr_max / divider / 1e4
        self.k = iteration                      # position inside the cycle
        self.rising = iteration < self.n_warm

    @staticmethod
    def _curve(shape, a, b, t):
        """value at fraction t of the way from a to b"""
        if shape == "linear":
            return a + t * (b - a)
        return b + (a - b) / 2 * (cos(pi * t) + 1)

    def lr_at(self, k, rising):
        if rising:
            return self._curve(self.shapes[0], self.lr_low, self.lr_max, k / self.n_warm)
        return self._curve(self.shapes[1], self.lr_max, self.lr_end, (k - self.n_warm) / (self.n_iter - self.n_warm))

    def step(self):
        self.k += 1
        lr = self.lr_at(self.k, self.rising)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
        if self.rising and self.k >= self.n_warm:
            self.rising = False
        elif not self.rising and self.k >= self.n_iter:
            self.k, self.rising = 0, True
        return lr


# The reference's util module also exports the pieces its scheduler class is assembled from (util.py:81-107): kept as
# importable names for code written as `from util import anneal_cosine, Phase`.  The scheduler above does not use them
# (it is a closed form of one counter); tests/test_host_cpu.py checks that both give the same learning rates.
def anneal_linear(start, end, proportion):
    return LinearWarmupCosineDecay._curve("linear", start, end, proportion)


def anneal_cosine(start, end, proportion):
    return LinearWarmupCosineDecay._curve("cosine", start, end, proportion)


class Phase:
    """One leg of a schedule: ``n_iter`` steps from ``start`` to ``end`` along ``anneal_fn`` (util.py:90-107)."""

    def __init__(self, start, end, n_iter, cur_iter, anneal_fn):
        self.start, self.end, self.n_iter, self.n, self.anneal_fn = start, end, n_iter, cur_iter, anneal_fn

    def step(self):
        self.n += 1
        return self.anneal_fn(self.start, self.end, self.n / self.n_iter)

    def reset(self):
        self.n = 0

    @property
    def is_done(self):
        return self.n >= self.n_iter


# ----------------------------------------------------------------------------- small host helpers
def flatten(v):
    """util.py:22-23: one level of nesting removed."""
    return [x for y in v for x in y]


def rescale(x):
    return (x - x.min()) / (x.max() - x.min())


def std_normal(size):
    """util.py:160-164: standard normal noise of ``size`` on the GPU."""
    return torch.randn(tuple(size), device="cuda")


def weight_scaling_init(layer):
    """util.py:168-175 (arXiv 1911.13254): weight and bias divided by sqrt(10 * std(weight)).  Written through ``.data``
    like the reference, which moves no version counter: the mutation epoch is bumped so that a cached eval artefact
    (TRUNet.folded) is rebuilt."""
    alpha = 10.0 * layer.weight.detach().std()
    layer.weight.data /= torch.sqrt(alpha)
    layer.bias.data /= torch.sqrt(alpha)
    L.bump_mutation_epoch()


def find_max_epoch(path):
    """util.py:30-49: largest <iter>.pkl in ``path`` or -1."""
    epoch = -1
    for f in os.listdir(path):
        if len(f) > 4 and f[-4:] == ".pkl":
            try:
                epoch = max(epoch, int(f[:-4]))
            except ValueError:
                continue
    return epoch


def print_size(net, keyword=None):
    """util.py:52-61."""
    if net is not None and isinstance(net, torch.nn.Module):
        params = sum(np.prod(p.size()) for p in net.parameters() if p.requires_grad)
        print("{} Parameters: {:.6f}M".format(net.__class__.__name__, params / 1e6), flush=True, end="; ")
        if keyword is not None:       # util.py:63-67: the share of the parameters whose name contains the keyword
            kp = sum(np.prod(p.size()) for n, p in net.named_parameters() if p.requires_grad and keyword in n)
            print("{} Parameters: {:.6f}M".format(keyword, kp / 1e6), flush=True, end="; ")
        print(" ")


@torch.no_grad()
def sampling(net, noisy_features):
    return net(noisy_features)
