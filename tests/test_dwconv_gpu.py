"""GPU: the fp32 depthwise-conv kernels (trunet_dwconv_fwd / trunet_dwconv_bwd: network.py:33-38 and its autograd) against a
torch fp64 restatement on random operands whose bias gradient does NOT vanish (the whole-network tests only see it next to a
BatchNorm, where it is analytically zero): outputs, statistics, data gradient, weight / bias gradient partial sums.  Shapes:
the three (kernel, stride) pairs of the encoder (sliding-window kernels) at lengths that do not divide into the position
chunks, with padding frames."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("K,S,Lin", [(3, 1, 13), (5, 2, 21), (3, 2, 16), (5, 2, 128), (3, 1, 64), (3, 2, 33)])
def test_dwconv_forward_backward_vs_fp64(K, S, Lin):
    from tinyrecurrentunet_amd import _lib as L
    from tinyrecurrentunet_amd._lib import check, ptr
    lib, st = L.lib(), L.stream()
    g = torch.Generator(device=DEV)
    g.manual_seed(100 * K + 10 * S + Lin)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV)
    N, NP, Cn = 1300, 1536, 24
    Lout = (Lin + 2 * (K // 2) - K) // S + 1
    zin = rnd(Cn, Lin, NP)
    sc, sh, mean = rnd(Cn) * 0.3 + 1, rnd(Cn) * 0.2, rnd(Cn) * 0.1
    wgt, b = rnd(Cn, 1, K) * 0.5, rnd(Cn) * 0.1
    zout = torch.empty(Cn, Lout, NP, device=DEV)
    nparts = lib.trunet_dwconv_nparts(Lout)
    part = torch.full((nparts * Cn * 2,), float("nan"), device=DEV)
    check(lib.trunet_dwconv_fwd(ptr(zin), ptr(sc), ptr(sh), ptr(wgt), ptr(b), ptr(zout), ptr(part), Cn, K, S, Lin, Lout, NP, N,
                                st), "fwd")
    act = torch.relu(sc[:, None, None] * zin + sh[:, None, None])
    ref = F.conv1d(act.permute(2, 0, 1).double(), wgt.double(), b.double(), stride=S, padding=K // 2, groups=Cn).permute(1, 2, 0)
    assert _l2(zout.double(), ref) < 1e-6
    r = ref[:, :, :N]
    st_ref = torch.stack([r.sum((1, 2)), (r * r).sum((1, 2))], 1)
    assert _l2(part.view(nparts, Cn, 2).double().sum(0), st_ref) < 1e-5

    dy, z = rnd(Cn, Lout, NP), rnd(Cn, Lout, NP)
    ca, cb, cc = rnd(Cn) * 0.5 + 1, rnd(Cn) * 0.1, rnd(Cn) * 0.05
    din = torch.empty(Cn, Lin, NP, device=DEV)
    nparts = lib.trunet_dwconv_bwd_nparts(Lin)
    part = torch.full((nparts * Cn * 2,), float("nan"), device=DEV)
    wp = torch.full((nparts * Cn * K,), float("nan"), device=DEV)
    bp = torch.full((nparts * Cn,), float("nan"), device=DEV)
    check(lib.trunet_dwconv_bwd(ptr(dy), ptr(z), ptr(ca), ptr(cb), ptr(cc), ptr(zin), ptr(sc), ptr(sh), ptr(mean), ptr(wgt),
                                ptr(din), ptr(part), ptr(wp), ptr(bp), Cn, K, S, Lin, Lout, NP, N, st), "bwd")
    dz = (ca[:, None, None] * dy + cb[:, None, None] * z + cc[:, None, None]).double()
    dz[:, :, N:] = 0
    actn = act.permute(2, 0, 1).double().requires_grad_(True)
    wd = wgt.double().requires_grad_(True)
    bd = b.double().requires_grad_(True)
    F.conv1d(actn, wd, bd, stride=S, padding=K // 2, groups=Cn).backward(dz.permute(2, 0, 1))
    gin = actn.grad.permute(1, 2, 0) * (act > 0)
    assert _l2(din.double(), gin) < 1e-6
    assert _l2(wp.view(nparts, -1).double().sum(0), wd.grad.reshape(-1)) < 1e-5
    assert _l2(bp.view(nparts, -1).double().sum(0), bd.grad) < 1e-5        # the bias gradient, not vanishing here
    r = gin[:, :, :N]
    st_ref = torch.stack([r.sum((1, 2)), (r * (zin[:, :, :N].double() - mean[:, None, None].double())).sum((1, 2))], 1)
    assert _l2(part.view(nparts, Cn, 2).double().sum(0), st_ref) < 1e-5


@pytest.mark.parametrize("K,S,Lin", [(3, 1, 13), (5, 2, 21), (3, 2, 16), (5, 2, 128), (3, 1, 64), (3, 2, 33), (3, 1, 2), (5, 2, 3)])
def test_dwconv_backward_with_recomputed_z(K, S, Lin, monkeypatch):
    """trunet_dwconv_bwd_rz (round 3): the backward that recomputes the conv's raw output z from its input rows instead of
    reading it.  With z = the forward kernel's own output both entry points must agree BIT FOR BIT on the data gradient and the
    weight / bias gradient sums (same dz, same order), chunk boundaries and ragged lengths included; the z-reading one is
    pinned against fp64 above.  One channel has a BatchNorm scale of exactly zero (the kernel's re-read path)."""
    from tinyrecurrentunet_amd import _lib as L
    from tinyrecurrentunet_amd._lib import check, ptr
    lib, st = L.lib(), L.stream()
    monkeypatch.setenv("TRUNET_DW_RZ", "2")      # also for fewer than 32 output positions (the entry point refuses those: slower)
    g = torch.Generator(device=DEV)
    g.manual_seed(7 + 100 * K + 10 * S + Lin)
    rnd = lambda *s: torch.randn(*s, generator=g, device=DEV)
    N, NP, Cn = 1300, 1536, 24
    Lout = (Lin + 2 * (K // 2) - K) // S + 1
    zin = rnd(Cn, Lin, NP)
    sc, sh, mean = rnd(Cn) * 0.3 + 1, rnd(Cn) * 0.2, rnd(Cn) * 0.1
    sc[5] = 0.0
    sh[5] = 0.3
    wgt, b = rnd(Cn, 1, K) * 0.5, rnd(Cn) * 0.1
    z = torch.empty(Cn, Lout, NP, device=DEV)
    part = torch.empty(lib.trunet_dwconv_nparts(Lout) * Cn * 2, device=DEV)
    check(lib.trunet_dwconv_fwd(ptr(zin), ptr(sc), ptr(sh), ptr(wgt), ptr(b), ptr(z), ptr(part), Cn, K, S, Lin, Lout, NP, N, st), "fwd")
    dy = rnd(Cn, Lout, NP)
    ca, cb, cc = rnd(Cn) * 0.5 + 1, rnd(Cn) * 0.1, rnd(Cn) * 0.05
    nparts = lib.trunet_dwconv_bwd_nparts(Lin)
    outs = []
    for rz in (False, True):
        din = torch.full((Cn, Lin, NP), float("nan"), device=DEV)
        pr = torch.full((nparts * Cn * 2,), float("nan"), device=DEV)
        wp = torch.full((nparts * Cn * K,), float("nan"), device=DEV)
        bp = torch.full((nparts * Cn,), float("nan"), device=DEV)
        tail = (ptr(ca), ptr(cb), ptr(cc), ptr(zin), ptr(sc), ptr(sh), ptr(mean), ptr(wgt), ptr(din), ptr(pr), ptr(wp), ptr(bp),
                Cn, K, S, Lin, Lout, NP, N, st)
        if rz:
            check(lib.trunet_dwconv_bwd_rz(ptr(dy), ptr(b), *tail), "bwd_rz")
        else:
            check(lib.trunet_dwconv_bwd(ptr(dy), ptr(z), *tail), "bwd")
        torch.cuda.synchronize()
        outs.append((din, pr, wp, bp))
    (din0, pr0, wp0, bp0), (din1, pr1, wp1, bp1) = outs
    assert torch.equal(din0, din1) and torch.equal(wp0, wp1) and torch.equal(bp0, bp1)
    # statistics: sum g bit for bit; sum g (zin - mean) takes zin - mean from the activation (a / sc - (sh / sc + mean) where
    # a > 0) instead of the raw row: equal to rounding
    pr0, pr1 = pr0.view(nparts, Cn, 2), pr1.view(nparts, Cn, 2)
    assert torch.equal(pr0[:, :, 0], pr1[:, :, 0])
    assert _l2(pr1[:, :, 1].double().sum(0), pr0[:, :, 1].double().sum(0)) < 1e-5


def test_dwconv_backward_rz_refuses_other_shapes():
    from tinyrecurrentunet_amd import _lib as L
    from tinyrecurrentunet_amd._lib import ptr
    lib, st = L.lib(), L.stream()
    t = torch.zeros(4 * 9 * 128, device=DEV)
    rc = lib.trunet_dwconv_bwd_rz(ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t),
                                  ptr(t), ptr(t), 4, 5, 1, 9, 9, 128, 100, st)
    assert rc == L.TRUNET_ENOTSUP
    # a supported (k, stride) with fewer than 32 output positions: refused as well (the z-reading kernel is faster there)
    rc = lib.trunet_dwconv_bwd_rz(ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t), ptr(t),
                                  ptr(t), ptr(t), 4, 3, 2, 9, 5, 128, 100, st)
    assert rc == L.TRUNET_ENOTSUP
