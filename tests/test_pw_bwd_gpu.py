"""GPU: the fused Conv1d(k=1) backward (trunet_pw_bwd) against the separate launches it replaces
(trunet_conv_wgrad + trunet_conv_gemm) on identical random operands, and against a torch fp64 restatement
of the same formulas.  Shapes are the ones the TRU-Net backward uses (network.py:28,50,64,83 + :96-98)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_case(N, P, M, srcs, left=0):
    """srcs: list of (nchan, L, kind) with kind in {"bn", "bn_accum", "relu_accum", "raw"}; the first source may be
    shifted by `left` positions (F.pad / crop of network.py:96-98)."""
    from tinyrecurrentunet_amd import _lib as L
    from tinyrecurrentunet_amd._lib import (DG_ACCUM, DG_MASK, DG_STATS, DG_STORE, EPI_ACCUM, EPI_MASK, EPI_STATS,
                                            PRO_BNBWD, PRO_BNRELU, PRO_NONE, GemmArgs, PwBwdArgs, WgradArgs, check,
                                            make_seg, ptr)
    lib = L.lib()
    st = L.stream()
    dev = "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(1000 * M + P + N)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
    NP = (N + 255) // 256 * 256
    K = sum(s[0] for s in srcs)
    dy, z = rnd(M, P, NP), rnd(M, P, NP)
    dy[:, :, N:] = 0
    ca, cb, cc = rnd(M) * 0.5 + 1, rnd(M) * 0.1, rnd(M) * 0.01
    W = rnd(M, K, 1) * 0.1
    tens, segs, woff = [], [], 0
    for i, (C, Ls, kind) in enumerate(srcs):
        zs = rnd(C, Ls, NP)
        sc, sh, mean = rnd(C) * 0.3 + 1, rnd(C) * 0.2, rnd(C) * 0.1
        if kind == "relu_accum":
            zs = torch.relu(zs)
        off = -left if (i == 0 and len(srcs) == 2) else 0
        if kind in ("bn", "bn_accum"):
            sg = make_seg(zs, C, Ls, pos_off=off, woff=woff, mode=PRO_BNRELU, c0=sc, c1=sh)
        else:
            sg = make_seg(zs, C, Ls, pos_off=off, woff=woff, mode=PRO_NONE)
        prev = rnd(C, Ls, NP) if "accum" in kind else None
        if prev is not None:
            prev[:, :, N:] = 0
        tens.append(dict(zs=zs, sc=sc, sh=sh, mean=mean, kind=kind, C=C, L=Ls, off=off, woff=woff, prev=prev))
        segs.append(sg)
        woff += C

    # ---------------- separate launches (the path the fused kernel replaces)
    npw = lib.trunet_conv_wgrad_nparts()
    a = WgradArgs()
    a.NP, a.N, a.P, a.p_begin = NP, N, P, 0
    a.M, a.a_L, a.a_pos_off, a.a_m_off = M, P, 0, 0
    a.ldw_m, a.ldw_c, a.w_m_off, a.nseg, a.w_numel = K, 1, 0, len(segs), W.numel()
    for i, sg in enumerate(segs):
        a.seg[i] = sg
    a.a0, a.a1, a.a_mode = ptr(dy), ptr(z), PRO_BNBWD
    a.ac0, a.ac1, a.ac2 = ptr(ca), ptr(cb), ptr(cc)
    wp = torch.zeros(npw * W.numel(), device=dev)
    bp = torch.zeros(npw * M, device=dev)
    a.w_partials, a.b_partials, a.b_stride, a.b_off = ptr(wp), ptr(bp), M, 0
    check(lib.trunet_conv_wgrad(a, st), "wgrad")
    gw_ref, gb_ref = torch.empty_like(W), torch.empty(M, device=dev)
    check(lib.trunet_reduce_partials(ptr(gw_ref), ptr(wp), npw, W.numel(), 0, st), "reduce")
    check(lib.trunet_reduce_partials(ptr(gb_ref), ptr(bp), npw, M, 0, st), "reduce")
    one, zero = torch.ones(256, device=dev), torch.zeros(256, device=dev)
    ref_out, ref_stats = [], []
    ngp = lib.trunet_conv_gemm_nparts(128)
    for t in tens:
        C, Ls, off = t["C"], t["L"], t["off"]
        p0, p1 = max(0, -off), min(P, Ls - off)
        out = t["prev"].clone() if t["prev"] is not None else torch.zeros(C, Ls, NP, device=dev)
        ga = GemmArgs()
        ga.NP, ga.N, ga.P, ga.p_begin = NP, N, p1 - p0, p0
        ga.M, ga.m_out_off, ga.out_L, ga.out_pos_off = C, 0, Ls, off
        ga.ldw_m, ga.ldw_c, ga.w_m_off, ga.nseg = 1, K, t["woff"], 1
        ga.seg[0] = make_seg(dy, M, P, mode=PRO_BNBWD, src1=z, c0=ca, c1=cb, c2=cc)
        ga.out, ga.W = ptr(out), ptr(W)
        epi = 0
        part = torch.zeros(ngp * C * 2, device=dev)
        if t["kind"] != "raw":
            epi |= EPI_MASK
            ga.zmask = ptr(t["zs"])
            bn = t["kind"].startswith("bn")
            ga.e0, ga.e1, ga.e2 = ptr(t["sc"] if bn else one), ptr(t["sh"] if bn else zero), ptr(t["mean"] if bn else zero)
            if "accum" in t["kind"]:
                epi |= EPI_ACCUM
            if bn:
                epi |= EPI_STATS
                ga.partials, ga.M_stat = ptr(part), C
        ga.epi = epi
        check(lib.trunet_conv_gemm(ga, st), "gemm")
        ref_out.append(out)
        ref_stats.append(part.view(ngp, C, 2).double().sum(0) if (epi & EPI_STATS) else None)

    # ---------------- fused launch
    f = PwBwdArgs()
    fw = f.w
    fw.NP, fw.N, fw.P, fw.p_begin = NP, N, P, 0
    fw.M, fw.a_L, fw.a_pos_off, fw.a_m_off = M, P, 0, 0
    fw.ldw_m, fw.ldw_c, fw.w_m_off, fw.nseg, fw.w_numel = K, 1, 0, len(segs), W.numel()
    fw.a0, fw.a1, fw.a_mode = ptr(dy), ptr(z), PRO_BNBWD
    fw.ac0, fw.ac1, fw.ac2 = ptr(ca), ptr(cb), ptr(cc)
    wp2 = torch.full((npw * W.numel(),), float("nan"), device=dev)
    bp2 = torch.full((npw * M,), float("nan"), device=dev)
    fw.w_partials, fw.b_partials, fw.b_stride, fw.b_off = ptr(wp2), ptr(bp2), M, 0
    f.W = ptr(W)
    nparts = lib.trunet_pw_bwd_nparts()
    outs, parts = [], []
    for i, (sg, t) in enumerate(zip(segs, tens)):
        fw.seg[i] = sg
        out = t["prev"].clone() if t["prev"] is not None else torch.zeros(t["C"], t["L"], NP, device=dev)
        part = torch.full((nparts * t["C"] * 2,), float("nan"), device=dev)
        d = f.dg[i]
        d.out = ptr(out)
        fl = DG_STORE
        if t["kind"] != "raw":
            fl |= DG_MASK
            d.zmask = ptr(t["zs"])
            if t["kind"].startswith("bn"):
                fl |= DG_STATS
                d.e2, d.partials = ptr(t["mean"]), ptr(part)
            if "accum" in t["kind"]:
                fl |= DG_ACCUM
        d.flags = fl
        outs.append(out)
        parts.append(part)
    check(lib.trunet_pw_bwd(f, st), "pw_bwd")
    gw, gb = torch.empty_like(W), torch.empty(M, device=dev)
    check(lib.trunet_reduce_partials(ptr(gw), ptr(wp2), npw, W.numel(), 0, st), "reduce")
    check(lib.trunet_reduce_partials(ptr(gb), ptr(bp2), npw, M, 0, st), "reduce")
    torch.cuda.synchronize()

    def rel(x, y):
        return float((x.double() - y.double()).norm() / (y.double().norm() + 1e-30))

    # ---------------- fp64 restatement of the formulas (independent of both kernels)
    dz = (ca.double()[:, None, None] * dy.double() + cb.double()[:, None, None] * z.double() + cc.double()[:, None, None])
    dz[:, :, N:] = 0
    W2 = W.double().view(M, K)
    for i, t in enumerate(tens):
        C, Ls, off = t["C"], t["L"], t["off"]
        p0, p1 = max(0, -off), min(P, Ls - off)
        g64 = torch.zeros(C, Ls, NP, dtype=torch.float64, device=dev)
        g64[:, p0 + off:p1 + off] = torch.einsum("mc,mpn->cpn", W2[:, t["woff"]:t["woff"] + C], dz[:, p0:p1])
        if t["prev"] is not None:
            g64 += t["prev"].double()
        if t["kind"] != "raw":
            zs = t["zs"].double()
            pre = zs * t["sc"].double()[:, None, None] + t["sh"].double()[:, None, None] if t["kind"].startswith("bn") else zs
            g64 = g64 * (pre > 0)
        e_f = rel(outs[i][..., :N], g64[..., :N])
        e_s = rel(ref_out[i][..., :N], g64[..., :N])
        assert e_f < 1e-5 and e_s < 1e-5, ("dgrad vs fp64: fused %.3g, separate %.3g" % (e_f, e_s), i)

    assert torch.isfinite(gw).all() and torch.isfinite(gb).all()
    assert rel(gw, gw_ref) < 1e-5, ("dW", rel(gw, gw_ref))
    assert rel(gb, gb_ref) < 1e-5, ("db", rel(gb, gb_ref))
    for i, t in enumerate(tens):
        assert torch.isfinite(outs[i]).all()
        # frames >= N are padding: the separate data-gradient kernel leaves W^T(cb z + cc) there, the fused one 0
        assert rel(outs[i][..., :N], ref_out[i][..., :N]) < 1e-6, ("dgrad", i, rel(outs[i][..., :N], ref_out[i][..., :N]))
        if ref_stats[i] is not None:
            s = parts[i].view(nparts, t["C"], 2).double().sum(0)
            scale = ref_stats[i].abs().max()
            assert float((s - ref_stats[i]).abs().max()) < 1e-4 * float(scale), ("stats", i)

    gw64 = torch.zeros(M, K, dtype=torch.float64, device=dev)
    for t in tens:
        C, Ls, off = t["C"], t["L"], t["off"]
        p0, p1 = max(0, -off), min(P, Ls - off)
        zs = t["zs"].double()
        if t["kind"].startswith("bn"):
            act = torch.relu(zs * t["sc"].double()[:, None, None] + t["sh"].double()[:, None, None])
        else:
            act = zs
        gw64[:, t["woff"]:t["woff"] + C] = torch.einsum("mpn,cpn->mc", dz[:, p0:p1], act[:, p0 + off:p1 + off])
    assert rel(gw.view(M, K), gw64) < 1e-4, rel(gw.view(M, K), gw64)
    assert rel(gb, dz.sum((1, 2))) < 1e-4


@pytest.mark.parametrize("N", [300, 1000])
def test_encoder_pw_128x128_accum(N):
    _run_case(N, 5, 128, [(128, 5, "bn_accum")])


def test_encoder_pw_128x64_relu_source():
    _run_case(300, 4, 128, [(64, 4, "relu_accum")])


def test_decoder_pw_two_sources_pad():
    # x1 shorter than the skip: F.pad by one on the left (decoder.1: 31 -> 32)
    _run_case(300, 8, 64, [(64, 7, "bn"), (128, 8, "raw")], left=1)


def test_decoder_pw_two_sources_crop():
    # x1 longer than the skip: crop one on each side (decoder.3: 66 -> 64)
    _run_case(520, 6, 64, [(64, 8, "bn"), (128, 6, "raw")], left=-1)


def test_fgru_conv_raw_source():
    _run_case(300, 16, 64, [(128, 16, "raw")])


def test_first_decoder_pw_single_bn_source():
    _run_case(700, 3, 64, [(64, 3, "bn")])


def _thin_wgrad_case(N, P, M, two, segspec):
    """trunet_conv_wgrad on the thin shapes (vector-ALU kernel) vs a torch fp64 restatement.
    segspec: list of (nchan, L, pos_mul, pos_off, pos_div, bn, shared_src_index or None)."""
    from tinyrecurrentunet_amd import _lib as L
    from tinyrecurrentunet_amd._lib import PRO_BNBWD, PRO_BNRELU, PRO_NONE, WgradArgs, check, make_seg, ptr
    lib, st, dev = L.lib(), L.stream(), "cuda"
    g = torch.Generator(device=dev)
    g.manual_seed(7 * M + P + N)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
    NP = (N + 255) // 256 * 256
    dy, z = rnd(M, P, NP), rnd(M, P, NP)
    ca, cb, cc = rnd(M) * 0.5 + 1, rnd(M) * 0.1, rnd(M) * 0.01
    srcs, segs, woff, info = {}, [], 0, []
    nseg = len(segspec)
    taps = all(s[6] is not None for s in segspec)
    for i, (C, Ls, pm, po, pdv, bn, share) in enumerate(segspec):
        key = share if share is not None else ("own", i)
        if key not in srcs:
            srcs[key] = (rnd(C, Ls, NP), rnd(C) * 0.3 + 1, rnd(C) * 0.2)
        t, sc, sh = srcs[key]
        wo = i if taps else woff
        segs.append(make_seg(t, C, Ls, pos_mul=pm, pos_off=po, pos_div=pdv, woff=wo,
                             mode=PRO_BNRELU if bn else PRO_NONE, c0=sc if bn else None, c1=sh if bn else None))
        info.append((t, sc, sh, C, Ls, pm, po, pdv, bn, wo))
        woff += C
    Ktot = woff
    # weight layouts: taps -> (M, C, k) conv style [ldw_m = C*k, ldw_c = k, woff = tap]; concat -> (M, Ktot)
    C0 = segspec[0][0]
    numel = M * (C0 * nseg if taps else Ktot)
    a = WgradArgs()
    a.NP, a.N, a.P, a.p_begin = NP, N, P, 0
    a.M, a.a_L, a.a_pos_off, a.a_m_off = M, P, 0, 0
    a.ldw_m, a.ldw_c = (C0 * nseg, nseg) if taps else (Ktot, 1)
    a.w_m_off, a.nseg, a.w_numel = 0, nseg, numel
    for i, sg in enumerate(segs):
        a.seg[i] = sg
    a.a0 = ptr(dy)
    if two:
        a.a1, a.a_mode, a.ac0, a.ac1, a.ac2 = ptr(z), PRO_BNBWD, ptr(ca), ptr(cb), ptr(cc)
    else:
        a.a_mode = PRO_NONE
    npw = lib.trunet_conv_wgrad_nparts()
    wp = torch.full((npw * numel,), float("nan"), device=dev)
    bp = torch.full((npw * M,), float("nan"), device=dev)
    a.w_partials, a.b_partials, a.b_stride, a.b_off = ptr(wp), ptr(bp), M, 0
    check(lib.trunet_conv_wgrad(a, st), "wgrad")
    gw, gb = torch.empty(numel, device=dev), torch.empty(M, device=dev)
    check(lib.trunet_reduce_partials(ptr(gw), ptr(wp), npw, numel, 0, st), "reduce")
    check(lib.trunet_reduce_partials(ptr(gb), ptr(bp), npw, M, 0, st), "reduce")
    torch.cuda.synchronize()
    dz = dy.double()
    if two:
        dz = ca.double()[:, None, None] * dz + cb.double()[:, None, None] * z.double() + cc.double()[:, None, None]
    dz = dz.clone()
    dz[:, :, N:] = 0
    ref = torch.zeros(numel, dtype=torch.float64, device=dev)
    for (t, sc, sh, C, Ls, pm, po, pdv, bn, wo) in info:
        act = t.double()
        if bn:
            act = torch.relu(act * sc.double()[:, None, None] + sh.double()[:, None, None])
        for p in range(P):
            qn = p * pm + po
            if qn < 0 or qn % pdv or qn // pdv >= Ls:
                continue
            contrib = dz[:, p] @ act[:, qn // pdv].T          # (M, C)
            idx = (torch.arange(M, device=dev)[:, None] * a.ldw_m + torch.arange(C, device=dev)[None, :] * a.ldw_c + wo)
            ref.index_put_((idx.reshape(-1),), contrib.reshape(-1), accumulate=True)
    err = float((gw.double() - ref).norm() / ref.norm())
    assert err < 1e-5, err
    errb = float((gb.double() - dz.sum((1, 2))).norm() / dz.sum((1, 2)).norm())
    assert errb < 1e-5, errb


@pytest.mark.parametrize("N", [300, 777])
def test_thin_wgrad_last_transposed_conv(N):
    # decoder.5 ConvTranspose1d(8, 8, 5, stride 2, padding 1): taps kk at q = (p + 1 - kk) / 2
    _thin_wgrad_case(N, 17, 8, False, [(8, 8, 1, 1 - kk, 2, True, "a") for kk in range(5)])


def test_thin_wgrad_first_conv():
    # encoder.0 Conv1d(4, 64, 5, stride 2, padding 1): taps kk at q = 2 p + kk - 1
    _thin_wgrad_case(300, 8, 64, False, [(4, 17, 2, kk - 1, 1, False, "x") for kk in range(5)])


def test_thin_wgrad_last_pointwise():
    # decoder.5 Conv1d(128 -> 8, k = 1) over [x1 | skip], BatchNorm backward on dz
    _thin_wgrad_case(520, 6, 8, True, [(64, 6, 1, 0, 1, True, None), (64, 6, 1, 0, 1, False, None)])


def test_thin_pointwise_backward_fused():
    # decoder.5 Conv1d(128 -> 8, k = 1) + BatchNorm over [x1 (BN, crop by one on each side) | skip (ReLU-only source, raw)]
    _run_case(520, 6, 8, [(64, 8, "bn"), (64, 6, "raw")], left=-1)
    _run_case(300, 5, 8, [(64, 5, "bn_accum"), (64, 5, "raw")])


def test_fused_backward_is_bitwise_repeatable(capsys):
    """Every kernel in the body is meant to be deterministic (two-stage reductions, no float atomics).  An earlier
    version of pw_bwd_kernel issued its epilogue loads as inline asm; under register pressure the compiler moved a
    not-yet-arrived value and about one launch in a hundred wrote a wrong 32x32 tile.  Repeat identical launches with
    cache / timing noise in between and compare every output bit for bit."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "dbg", "stress_pwbwd.py")
    spec = importlib.util.spec_from_file_location("stress_pwbwd", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.run(300, 64, 128, [(128, "bn_accum")], 150, "enc", noise=True)
    mod.run(300, 32, 64, [(64, "bn"), (128, "raw")], 100, "dec", noise=True)
    out = capsys.readouterr().out
    assert "enc: 0 bad of 150" in out and "dec: 0 bad of 100" in out, out
