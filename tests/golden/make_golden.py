#!/usr/bin/env python3
"""Generate golden vectors from the REFERENCE's own importable pieces (build container only).

Run from the repo root:  python tests/golden/make_golden.py
Reads /root/reference at run time (never copied): ``network.py`` lines 1-121 (the six block
classes; the ``TRUNet`` class below them is a SyntaxError), ``stft_loss.py``,
``dataset.py`` (with the absent ``librosa``/``torchaudio`` imports stubbed -- they are used only by
the data-loading code), and ``util.py`` lines 81-156 (LR schedule).  Writes small ``.npz``
fixtures next to this script.  Weights come from ``oracle.weights`` (numpy PCG64 by tensor
name), so only inputs/outputs are stored.
"""
import importlib
import math
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")

from oracle import weights as W  # noqa: E402


def load_reference():
    sys.path.insert(0, REF)
    ref = types.SimpleNamespace()
    ref.phm = importlib.import_module("phm")
    src = open(os.path.join(REF, "network.py")).read().split("\n")
    net = types.ModuleType("ref_network_blocks")
    exec(compile("\n".join(src[:121]), "ref_network_blocks", "exec"), net.__dict__)
    ref.net = net
    ref.stft_loss = importlib.import_module("stft_loss")
    for n in ["librosa", "torchaudio", "torchaudio.transforms", "torchaudio.functional"]:
        sys.modules.setdefault(n, types.ModuleType(n))
    ta = sys.modules["torchaudio"]
    ta.transforms = sys.modules["torchaudio.transforms"]
    ta.functional = sys.modules["torchaudio.functional"]

    class _Stub:
        def __init__(self, *a, **k):
            pass
    ta.transforms.Spectrogram = _Stub
    ta.transforms.InverseSpectrogram = _Stub
    ref.dataset = importlib.import_module("dataset")
    usrc = open(os.path.join(REF, "util.py")).read().split("\n")
    sched = types.ModuleType("ref_sched")
    sched.__dict__.update(cos=math.cos, pi=math.pi)
    exec(compile("\n".join(usrc[80:156]), "ref_sched", "exec"), sched.__dict__)
    ref.sched = sched
    return ref


class RefComposition(torch.nn.Module):
    """R1-R4 composition built from the REFERENCE's block classes."""

    def __init__(self, rn, c_in):
        super().__init__()
        nn = torch.nn
        self.encoder = nn.ModuleList([rn.StandardConv1d(c_in, 64, 5, 2),
                                      rn.DepthwiseSeparableConv1d(64, 128, 3, 1),
                                      rn.DepthwiseSeparableConv1d(128, 128, 5, 2),
                                      rn.DepthwiseSeparableConv1d(128, 128, 3, 1),
                                      rn.DepthwiseSeparableConv1d(128, 128, 5, 2),
                                      rn.DepthwiseSeparableConv1d(128, 128, 3, 2)])
        self.decoder = nn.ModuleList([rn.FirstTrCNN(64, 64, 3, 2), rn.TrCNN(192, 64, 5, 2),
                                      rn.TrCNN(192, 64, 3, 1), rn.TrCNN(192, 64, 5, 2),
                                      rn.TrCNN(192, 64, 3, 1), rn.LastTrCNN(128, 8, 5, 2)])
        self.FGRU = rn.GRUBlock(128, 64, 64, bidirectional=True)
        self.TGRU = rn.GRUBlock(64, 128, 64, bidirectional=False)

    def forward(self, x):
        skips = []
        for blk in self.encoder:
            x = blk(x)
            skips.append(x)
        skips = skips[::-1]
        x = self.FGRU(x.transpose(1, 2))
        x = self.decoder[0](x)
        for i in range(1, 6):
            x = self.decoder[i](x, skips[i])
        return x


def rnd(shape, seed, scale=1.0):
    return torch.tensor(np.random.default_rng(seed).standard_normal(shape) * scale, dtype=torch.float32)


def save(name, **arrs):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()}
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote %-28s %7.1f KB" % (name, os.path.getsize(os.path.join(HERE, name + ".npz")) / 1024))


def gen_blocks(ref, only=None):
    rn = ref.net
    specs = {
        # the unidirectional class of network.py:150 (TGRU), stand-alone: 3 sequences of 7 steps
        "gru_uni": (rn.GRUBlock(64, 128, 64, False), [(3, 7, 64)]),
        "std": (rn.StandardConv1d(4, 64, 5, 2), [(2, 4, 257)]),
        "dsc_k3s1": (rn.DepthwiseSeparableConv1d(64, 128, 3, 1), [(2, 64, 32)]),
        "dsc_k5s2": (rn.DepthwiseSeparableConv1d(128, 128, 5, 2), [(2, 128, 32)]),
        "dsc_k3s2": (rn.DepthwiseSeparableConv1d(128, 128, 3, 2), [(2, 128, 32)]),
        "gru_bi": (rn.GRUBlock(128, 64, 64, True), [(3, 16, 128)]),
        "first_tr": (rn.FirstTrCNN(64, 64, 3, 2), [(2, 64, 16)]),
        "tr_k5s2": (rn.TrCNN(192, 64, 5, 2), [(2, 64, 31), (2, 128, 32)]),
        "tr_k3s1": (rn.TrCNN(192, 64, 3, 1), [(2, 64, 65), (2, 128, 64)]),
        "last_tr": (rn.LastTrCNN(128, 8, 5, 2), [(2, 64, 130), (2, 64, 128)]),
    }
    for name, (mod, shapes) in specs.items():
        if only and name not in only:
            continue
        W.fill_state_dict(mod, seed=11)
        wsum = W.checksum(mod)
        ins = [rnd(s, 100 + i) for i, s in enumerate(shapes)]
        mod.eval()
        with torch.no_grad():
            y_eval = mod(*[t.clone() for t in ins])
        mod.train()
        xs = [t.clone().requires_grad_(True) for t in ins]
        y = mod(*xs)
        cot = rnd(tuple(y.shape), 7)
        (y * cot).sum().backward()
        arrs = {"y_eval": y_eval, "y_train": y, "cot": cot, "wsum": wsum}
        for i, (t, x) in enumerate(zip(ins, xs)):
            arrs["x%d" % i] = t
            arrs["gx%d" % i] = x.grad
        for pn, p in mod.named_parameters():
            arrs["g:" + pn] = p.grad
        for bn, b in mod.named_buffers():
            if b.is_floating_point():
                arrs["buf:" + bn] = b
        save("block_" + name, **arrs)


def gen_composition(ref):
    for c_in in (3, 4):
        net = RefComposition(ref.net, c_in)
        W.fill_state_dict(net, seed=0)
        wsum = W.checksum(net)
        x = rnd((5, c_in, 257), 42 + c_in, 0.7)
        net.eval()
        with torch.no_grad():
            y_eval = net(x.clone())
        net.train()
        y = net(x.clone())
        cot = rnd(tuple(y.shape), 9)
        (y * cot).sum().backward()
        arrs = {"x": x, "y_eval": y_eval, "y_train": y, "cot": cot, "wsum": wsum,
                "n_grad_params": sum(p.numel() for p in net.parameters() if p.grad is not None)}
        for pn, p in net.named_parameters():
            if p.grad is not None:
                arrs["g:" + pn] = p.grad
        for bn, b in net.named_buffers():
            if b.is_floating_point() and "TGRU" not in bn:
                arrs["buf:" + bn] = b
        save("trunet_cin%d" % c_in, **arrs)


def gen_stft_loss(ref):
    cfg = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200],
               sc_lambda=0.5, mag_lambda=0.5, band="full")
    m = ref.stft_loss.MultiResolutionSTFTLoss(**cfg)
    x = rnd((2, 4000), 1, 0.1).requires_grad_(True)
    y = rnd((2, 4000), 2, 0.1)
    sc, mag = m(x, y)
    (sc + mag).backward()
    per = []
    for f in m.stft_losses:
        a, b = f(x.detach(), y)
        per += [float(a), float(b)]
    # survey known answer: torch.manual_seed(0) randn pair at (2,16000)
    torch.manual_seed(0)
    xs = torch.randn(2, 16000)
    ys = torch.randn(2, 16000)
    ksc, kmag = m(xs, ys)
    save("mrstft", x=x, y=y, sc=sc, mag=mag, gx=x.grad, per_res=np.array(per),
         known_sc=ksc, known_mag=kmag)


def gen_features(ref):
    pa = ref.dataset.ProcessAudio()
    clean, noisy = W.synth_pairs(1, 2048, seed=5)
    feat = pa(noisy.clone())                       # (17,3,257)
    back = pa.backward(feat.clone())               # (1,2048)
    mag = torch.stft(noisy[0], 512, 128, return_complex=True).abs()   # (1,257,17)
    pc = ref.dataset.pcenfunc(mag.transpose(1, 2).clone(), training=True)
    pc_eval = ref.dataset.pcenfunc(mag.transpose(1, 2).clone(), training=False)
    # arbitrary (non-unit-modulus) features through the inverse path
    f2 = rnd((9, 3, 257), 3, 0.6)
    back2 = pa.backward(f2.clone())
    save("features", audio=noisy, feat=feat, back=back, pcen_train=pc, pcen_eval=pc_eval,
         feat2=f2, back2=back2)


def gen_sched(ref):
    class _Opt:
        param_groups = [{"lr": 0.0}]
    s = ref.sched.LinearWarmupCosineDecay(_Opt(), lr_max=4e-4, n_iter=1000, iteration=0, divider=25,
                                          warmup_proportion=0.05, phase=("linear", "cosine"))
    lrs = [s.step() for _ in range(1000)]
    save("sched", lrs=np.array(lrs, dtype=np.float64))


def gen_sched_wrap(ref):
    """the same class across its wrap-around (two and a bit cycles of 40 steps) and resumed inside either phase"""
    class _Opt:
        param_groups = [{"lr": 0.0}]
    mk = lambda it: ref.sched.LinearWarmupCosineDecay(_Opt(), lr_max=4e-4, n_iter=40, iteration=it, divider=25,  # noqa: E731
                                                      warmup_proportion=0.25, phase=("linear", "cosine"))
    s = mk(0)
    cyc = [s.step() for _ in range(95)]
    s = mk(4)
    res_warm = [s.step() for _ in range(50)]
    s = mk(25)
    res_decay = [s.step() for _ in range(50)]
    s = ref.sched.LinearWarmupCosineDecay(_Opt(), lr_max=1e-3, n_iter=30, iteration=0, divider=10, warmup_proportion=0.3,
                                          phase=("cosine", "linear"))
    swapped = [s.step() for _ in range(70)]
    save("sched_wrap", cycles=np.array(cyc), resumed_warm=np.array(res_warm), resumed_decay=np.array(res_decay),
         swapped=np.array(swapped))


def gen_stft_fn(ref):
    """the stand-alone stft() of stft_loss.py:9-30 with its autograd gradient; one input row is silent over a stretch so
    that clamp(min=1e-7) is active for whole frames"""
    x = rnd((2, 2000), 31, 0.1)
    x[1, 600:1500] = 0.0
    outs = {"x": x}
    for n, hop, wl in ((512, 120, 240), (1024, 250, 600)):
        xx = x.clone().requires_grad_(True)
        mag = ref.stft_loss.stft(xx, n, hop, wl, torch.hann_window(wl))
        cot = rnd(tuple(mag.shape), 32 + n)
        (mag * cot).sum().backward()
        outs.update({"mag_%d" % n: mag, "cot_%d" % n: cot, "gx_%d" % n: xx.grad})
    save("stft_fn", **outs)


def gen_phm(ref):
    """PhaseAwareMask.forward of phm.py:31-45, UNMODIFIED.  Its line 41 reads two names, `phase_mix` and `phase_est`,
    that the function never binds (it binds `phase_mixture` / `phase_estimated`), so as written it raises
    NameError; Python resolves them as module globals, so binding those two globals to the angles the function itself
    computes two lines above makes the reference's own code produce the value it evidently means (repair R5)."""
    rng = np.random.default_rng(77)
    shape = (257, 21)
    mix = torch.complex(torch.tensor(rng.standard_normal(shape), dtype=torch.float32),
                        torch.tensor(rng.standard_normal(shape), dtype=torch.float32))
    est = torch.complex(torch.tensor(rng.standard_normal(shape), dtype=torch.float32),
                        torch.tensor(rng.standard_normal(shape), dtype=torch.float32))
    # edge cases of angle()/abs(): zeros, purely real negative (angle = pi), purely imaginary
    mix[0, 0] = 0
    est[0, 1] = 0
    mix[1, 0] = -1.5
    est[1, 1] = -0.25
    mix[2, 0] = 2.0j
    est[2, 1] = -3.0j
    outs = {}
    for beta in (0.5, 2.0):
        m = mix.clone().requires_grad_(True)
        e = est.clone().requires_grad_(True)
        ref.phm.phase_mix = torch.angle(m)
        ref.phm.phase_est = torch.angle(e)
        out = ref.phm.PhaseAwareMask(beta).forward(m, e)
        cot = torch.tensor(np.random.default_rng(78).standard_normal(shape), dtype=torch.float32)
        (out * cot).sum().backward()
        tag = "b%02d" % int(beta * 10)
        outs.update({"out_" + tag: out, "gmix_" + tag: torch.view_as_real(m.grad), "gest_" + tag: torch.view_as_real(e.grad)})
        del ref.phm.phase_mix, ref.phm.phase_est
    save("phm", mix=torch.view_as_real(mix), est=torch.view_as_real(est), cot=cot, **outs)


def gen_config():
    """the values of config/tiny.json's sections that reach the hot-path modules as **kwargs (train.py:180-192): data
    for the drop-in tests (constructor / loss_fn / loader keyword names and values), no code"""
    import json
    cfg = json.load(open(os.path.join(REF, "config", "tiny.json")))
    out = {"network": cfg["network"], "loss_config": cfg["train"]["loss_config"],
           "optimization": cfg["train"]["optimization"],
           "trainset": {k: v for k, v in cfg["trainset"].items() if k != "root"}, "dist": cfg["dist"]}
    with open(os.path.join(HERE, "tiny_config_sections.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote tiny_config_sections.json")


if __name__ == "__main__":
    torch.manual_seed(0)
    if sys.argv[1:] == ["config"]:
        gen_config()
        sys.exit(0)
    ref = load_reference()
    if len(sys.argv) > 1:               # python tests/golden/make_golden.py block:gru_uni phm ...  (add single fixtures)
        only = [a.split(":", 1)[1] for a in sys.argv[1:] if a.startswith("block:")]
        if only:
            gen_blocks(ref, only=only)
        if "phm" in sys.argv[1:]:
            gen_phm(ref)
        if "sched_wrap" in sys.argv[1:]:
            gen_sched_wrap(ref)
        if "stft_fn" in sys.argv[1:]:
            gen_stft_fn(ref)
        sys.exit(0)
    gen_config()
    gen_blocks(ref)
    gen_composition(ref)
    gen_stft_loss(ref)
    gen_features(ref)
    gen_sched(ref)
    gen_sched_wrap(ref)
    gen_phm(ref)
    gen_stft_fn(ref)
