"""CPU: the oracle restatement vs golden vectors produced by the reference's own pieces
(tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from oracle import features_ref as fr
from oracle import loss_ref, network_ref as nr, stft_loss_ref as sl, weights as W

T = torch.tensor
BLOCKS = {
    "std": lambda: nr.StandardConv1d(4, 64, 5, 2),
    "dsc_k3s1": lambda: nr.DepthwiseSeparableConv1d(64, 128, 3, 1),
    "dsc_k5s2": lambda: nr.DepthwiseSeparableConv1d(128, 128, 5, 2),
    "dsc_k3s2": lambda: nr.DepthwiseSeparableConv1d(128, 128, 3, 2),
    "gru_bi": lambda: nr.GRUBlock(128, 64, 64, True),
    "gru_uni": lambda: nr.GRUBlock(64, 128, 64, False),
    "first_tr": lambda: nr.FirstTrCNN(64, 64, 3, 2),
    "tr_k5s2": lambda: nr.TrCNN(192, 64, 5, 2),
    "tr_k3s1": lambda: nr.TrCNN(192, 64, 3, 1),
    "last_tr": lambda: nr.LastTrCNN(128, 8, 5, 2),
}


def close(a, b, rtol=1e-5, atol=1e-6):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", sorted(BLOCKS))
def test_block_matches_reference(golden, name):
    g = golden("block_" + name)
    mod = W.fill_state_dict(BLOCKS[name](), seed=11)
    assert abs(W.checksum(mod) - float(g["wsum"])) < 1e-6 * float(g["wsum"]), "weight RNG drift"
    ins = [T(g["x%d" % i]) for i in range(2) if "x%d" % i in g]
    mod.eval()
    with torch.no_grad():
        close(mod(*[t.clone() for t in ins]), g["y_eval"])
    mod.train()
    xs = [t.clone().requires_grad_(True) for t in ins]
    y = mod(*xs)
    close(y, g["y_train"])
    (y * T(g["cot"])).sum().backward()
    for i, x in enumerate(xs):
        close(x.grad, g["gx%d" % i], rtol=1e-4, atol=1e-5)
    for pn, p in mod.named_parameters():
        close(p.grad, g["g:" + pn], rtol=1e-4, atol=1e-4)
    for bn, b in mod.named_buffers():
        if b.is_floating_point():
            close(b, g["buf:" + bn])


@pytest.mark.parametrize("c_in", [3, 4])
def test_trunet_composition_matches_reference_blocks(golden, c_in):
    g = golden("trunet_cin%d" % c_in)
    net = W.fill_state_dict(nr.TRUNet(input_size=c_in), seed=0)
    assert len(net.state_dict()) == 177
    assert abs(W.checksum(net) - float(g["wsum"])) < 1e-6 * float(g["wsum"])
    x = T(g["x"])
    net.eval()
    with torch.no_grad():
        close(net(x.clone()), g["y_eval"], rtol=1e-4, atol=1e-5)
    net.train()
    y = net(x.clone())
    close(y, g["y_train"], rtol=1e-4, atol=1e-5)
    (y * T(g["cot"])).sum().backward()
    n = 0
    for pn, p in net.named_parameters():
        if p.grad is None:
            assert "TGRU" in pn
            continue
        n += p.numel()
        close(p.grad, g["g:" + pn], rtol=1e-3, atol=1e-3)
    assert n == int(g["n_grad_params"]) == (298272 if c_in == 3 else 298592)


def test_mrstft_matches_reference(golden):
    g = golden("mrstft")
    x = T(g["x"]).requires_grad_(True)
    sc, mag = sl.mr_stft_loss(x, T(g["y"]))
    close(sc, g["sc"], rtol=1e-5)
    close(mag, g["mag"], rtol=1e-5)
    (sc + mag).backward()
    close(x.grad, g["gx"], rtol=1e-4, atol=1e-7)
    torch.manual_seed(0)
    xs, ys = torch.randn(2, 16000), torch.randn(2, 16000)
    ksc, kmag = sl.mr_stft_loss(xs, ys)
    close(ksc, g["known_sc"], rtol=1e-5)
    close(kmag, g["known_mag"], rtol=1e-5)
    # SURVEY 8c known answers
    assert abs(float(ksc) - 0.33027813) < 1e-5 and abs(float(kmag) - 0.35075721) < 1e-5


def test_features_match_reference(golden):
    g = golden("features")
    audio = T(g["audio"])
    feat = fr.features_one(audio)
    close(feat, g["feat"], rtol=1e-5, atol=2e-6)
    close(fr.inverse_features_one(T(g["feat"])), g["back"], rtol=1e-5, atol=1e-6)
    close(fr.inverse_features_one(T(g["feat2"])), g["back2"], rtol=1e-4, atol=1e-5)
    mag = fr.stft_rect(audio[0]).abs().t().unsqueeze(0)
    close(fr.pcen_ref(mag), g["pcen_train"], rtol=1e-5, atol=1e-6)
    close(fr.pcen_ref(mag), g["pcen_eval"], rtol=1e-5, atol=1e-6)
    f4 = fr.features_one(audio, pcen=True)
    assert f4.shape == (17, 4, 257)
    close(f4[:, 1], g["pcen_train"][0], rtol=1e-5, atol=1e-6)
    close(f4[:, [0, 2, 3]], g["feat"], rtol=1e-5, atol=2e-6)


def test_phm_closed_form():
    torch.manual_seed(3)
    m = torch.randn(257, 9, dtype=torch.cfloat)
    e = torch.randn(257, 9, dtype=torch.cfloat)
    out = fr.phase_aware_mask(m, e, beta=0.5)
    ref = torch.sigmoid(0.5 * (torch.angle(m) - torch.angle(e))) * m.abs()
    close(out, ref.numpy(), rtol=1e-6)


def test_sched_matches_reference(golden):
    lrs = golden("sched")["lrs"]
    for s in (1, 2, 49, 50, 51, 300, 525, 999, 1000):
        assert abs(loss_ref.lr_schedule(s, 4e-4, 1000, 25, 0.05) - lrs[s - 1]) < 1e-15 + 1e-12 * lrs[s - 1]
    for s, v in ((1, 2.368e-05), (50, 4e-4), (51, 3.999989e-4), (525, 2.000008e-4), (1000, 1.6e-9)):
        assert abs(lrs[s - 1] - v) < 2e-10


def test_phm_matches_reference(golden):
    """a10: the oracle's mask against values the REFERENCE's own unmodified PhaseAwareMask.forward produced (its two
    unbound names bound as module globals by make_golden.gen_phm), forward and autograd gradients, incl. the
    angle()/abs() edge cases (zeros, negative real, purely imaginary inputs)"""
    g = golden("phm")
    for tag, beta in (("b05", 0.5), ("b20", 2.0)):
        m = torch.view_as_complex(T(g["mix"])).clone().requires_grad_(True)
        e = torch.view_as_complex(T(g["est"])).clone().requires_grad_(True)
        out = fr.phase_aware_mask(m, e, beta=beta)
        close(out, g["out_" + tag], rtol=1e-6, atol=1e-7)
        (out * T(g["cot"])).sum().backward()
        close(torch.view_as_real(m.grad), g["gmix_" + tag], rtol=1e-5, atol=1e-6)
        close(torch.view_as_real(e.grad), g["gest_" + tag], rtol=1e-5, atol=1e-6)


def test_stft_fn_matches_reference(golden):
    """a12: stft() magnitudes and their autograd gradient vs the reference's stft_loss.stft"""
    g = golden("stft_fn")
    for n, hop, wl in ((512, 120, 240), (1024, 250, 600)):
        x = T(g["x"]).clone().requires_grad_(True)
        mag = sl.stft_mag(x, n, hop, wl, torch.hann_window(wl))
        close(mag, g["mag_%d" % n], rtol=1e-5, atol=1e-6)
        (mag * T(g["cot_%d" % n])).sum().backward()
        close(x.grad, g["gx_%d" % n], rtol=1e-4, atol=1e-5)


def test_sched_wrap_matches_reference(golden):
    """the product's closed-form LinearWarmupCosineDecay across wrap-arounds, resumed in either phase, and with the two
    curve shapes swapped, vs the reference class (util.py:110-156)"""
    from tinyrecurrentunet_amd import util
    g = golden("sched_wrap")

    class Opt:
        param_groups = [{"lr": 0.0}]

    def run(n, **kw):
        s = util.LinearWarmupCosineDecay(Opt(), **kw)
        return np.array([s.step() for _ in range(n)])
    base = dict(lr_max=4e-4, n_iter=40, divider=25, warmup_proportion=0.25, phase=("linear", "cosine"))
    np.testing.assert_allclose(run(95, iteration=0, **base), g["cycles"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(run(50, iteration=4, **base), g["resumed_warm"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(run(50, iteration=25, **base), g["resumed_decay"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(run(70, lr_max=1e-3, n_iter=30, iteration=0, divider=10, warmup_proportion=0.3,
                                   phase=("cosine", "linear")), g["swapped"], rtol=1e-12, atol=0)
