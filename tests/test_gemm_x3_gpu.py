"""GPU: conv_gemm_x3_kernel (gemm_x3.hip) -- the forward implicit GEMM on the bf16 MFMA with an exact three-term split of the
fp32 operands -- against the fp32-MFMA kernel it replaces (same entry point, trunet_gemm_x3_enable(0)) and against a float64
product of the same fp32 operands.  The claim under test: it IS an fp32 GEMM (24 significant bits per product, fp32
accumulation), so its error against float64 must be of the size of the fp32-MFMA kernel's own."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ref(W, ldw_m, ldw_c, w_m_off, M, segs, bias, P, out_L, out_pos_off, NP):
    """float64: out[m][p + off][n] = bias[m] + sum_seg sum_c W[m + w_m_off, c, seg] * pro(src[c][q(p)][n])"""
    out = torch.zeros(M, out_L, NP, dtype=torch.float64, device=DEV)
    Wf = W.double().reshape(-1)
    for p in range(P):
        acc = torch.zeros(M, NP, dtype=torch.float64, device=DEV)
        for s in segs:
            qn = p * s["mul"] + s["off"]
            if qn < 0 or qn % s["div"] or qn // s["div"] >= s["x"].shape[1]:
                continue
            v = s["x"][:, qn // s["div"]].double()
            if s.get("c0") is not None:
                v = torch.relu(s["c0"].double()[:, None] * v + s["c1"].double()[:, None])
            C = v.shape[0]
            idx = (torch.arange(M, device=DEV)[:, None] + w_m_off) * ldw_m + torch.arange(C, device=DEV)[None, :] * ldw_c + s["woff"]
            acc += Wf[idx] @ v
        out[:, p + out_pos_off] = acc + (bias.double()[:, None] if bias is not None else 0.0)
    return out


def _run(M, segs, W, ldw_m, ldw_c, P, N, NP, bias=True, w_m_off=0, out_L=None, out_pos_off=0):
    from tinyrecurrentunet_amd import _lib as L
    from tinyrecurrentunet_amd._lib import PRO_BNRELU, PRO_NONE, make_seg
    from tinyrecurrentunet_amd.engine import TRUNetEngine, Workspace
    import ctypes as C
    out_L = out_L or P
    b = (torch.randn(M, device=DEV) * 0.3) if bias else None
    sg = lambda: [make_seg(s["x"], s["x"].shape[0], s["x"].shape[1], s["mul"], s["off"], s["div"], s["woff"],
                           PRO_BNRELU if s.get("c0") is not None else PRO_NONE, c0=s.get("c0"), c1=s.get("c1")) for s in segs]
    res = {}
    lib = L.lib()
    prev = lib.trunet_gemm_x3_enable(-1)
    try:
        for mode in (1, 0):
            lib.trunet_gemm_x3_enable(mode)
            eng, w = TRUNetEngine(None), Workspace(torch.device(DEV))
            out = torch.full((M, out_L, NP), float("nan"), device=DEV)
            a = eng._gemm_args(N=N, NP=NP, P=P, M=M, out=out, out_L=out_L, W=W, ldw_m=ldw_m, ldw_c=ldw_c, segs=sg(), bias=b)
            v = [C.c_int() for _ in range(6)]
            L.check(lib.trunet_conv_gemm_plan(a, *[C.byref(x) for x in v]), "plan")
            assert (v[5].value == -8) == (mode == 1), "x3 %s but the plan says nw = %d" % ("on" if mode else "off", v[5].value)
            nparts = eng._gemm(w, N=N, NP=NP, P=P, M=M, out=out, out_L=out_L, W=W, ldw_m=ldw_m, ldw_c=ldw_c, segs=sg(),
                               bias=b, w_m_off=w_m_off, out_pos_off=out_pos_off, stats=M)
            torch.cuda.synchronize()
            st = w.t["partials"][:nparts * M * 2].view(nparts, M, 2).double().sum(0)
            res[mode] = (out.clone(), st)
    finally:
        lib.trunet_gemm_x3_enable(prev)
    ref = _ref(W, ldw_m, ldw_c, w_m_off, M, segs, b, P, out_L, out_pos_off, NP)
    rows = slice(out_pos_off, out_pos_off + P)
    scale = ref[:, rows].abs().max().item()
    e3 = (res[1][0][:, rows].double() - ref[:, rows]).abs().max().item() / scale
    e1 = (res[0][0][:, rows].double() - ref[:, rows]).abs().max().item() / scale
    l3 = ((res[1][0][:, rows].double() - ref[:, rows]).norm() / ref[:, rows].norm()).item()
    l1 = ((res[0][0][:, rows].double() - ref[:, rows]).norm() / ref[:, rows].norm()).item()
    print("M%d K%s P%d: vs float64 max-abs / max|out|: x3 %.2e fp32-MFMA %.2e; relative L2: x3 %.2e fp32-MFMA %.2e" % (
        M, "+".join(str(s["x"].shape[0]) for s in segs), P, e3, e1, l3, l1))
    assert torch.isfinite(res[1][0][:, rows]).all()
    assert l3 < 2.0 * l1 + 1e-7 and e3 < 3.0 * e1 + 2e-7, (e3, e1, l3, l1)
    # statistics (sum, sum of squares over the N valid frames) of both kernels agree with the float64 ones
    r = ref[:, rows, :N]
    st_ref = torch.stack([r.sum((1, 2)), (r * r).sum((1, 2))], 1)
    for mode in (1, 0):
        d = (res[mode][1] - st_ref).abs().max().item() / st_ref.abs().max().item()
        assert d < 2e-5, (mode, d)
    return res


def _x(C, Ln, NP, N, g):
    x = torch.randn(C, Ln, NP, generator=g, device=DEV)
    x[:, :, N:] = 0.0                          # padding frames: finite
    return x


@pytest.mark.parametrize("M,K,P,N", [(128, 128, 7, 700), (128, 64, 5, 257), (64, 64, 4, 1000), (128, 128, 40, 8200),
                                     (64, 128, 33, 12000)])
def test_x3_pointwise_one_source(M, K, P, N):
    """(the last two: 1,320 / 1,551 tiles, several per persistent workgroup -- the software pipeline runs ACROSS tile
    boundaries: epilogue stores in flight next to the ring's DMA, accumulators re-zeroed, relaxed waits)"""
    g = torch.Generator(device=DEV).manual_seed(M + K + P)
    NP = (N + 255) // 256 * 256
    W = torch.randn(M, K, generator=g, device=DEV) * 0.2
    seg = dict(x=_x(K, P, NP, N, g), mul=1, off=0, div=1, woff=0, c0=torch.rand(K, generator=g, device=DEV) + 0.5,
               c1=torch.randn(K, generator=g, device=DEV) * 0.3)
    _run(M, [seg], W, K, 1, P, N, NP)


def test_x3_decoder_pointwise_two_sources_with_pad():
    """decoder pointwise conv over [x1 shifted by F.pad | skip] (network.py:95-100): M = 64, K = 64 + 128"""
    g = torch.Generator(device=DEV).manual_seed(5)
    N, NP, P = 600, 768, 8
    W = torch.randn(64, 192, generator=g, device=DEV) * 0.15
    mk = lambda C, Ln: dict(x=_x(C, Ln, NP, N, g), mul=1, div=1, c0=torch.rand(C, generator=g, device=DEV) + 0.5,
                            c1=torch.randn(C, generator=g, device=DEV) * 0.3)
    s1 = dict(mk(64, 7), off=-1, woff=0)               # one position of left padding
    s2 = dict(mk(128, 8), off=0, woff=64)
    _run(64, [s1, s2], W, 192, 1, P, N, NP)


@pytest.mark.parametrize("k,s", [(3, 1), (5, 2), (3, 2)])
def test_x3_transposed_conv_taps(k, s):
    """ConvTranspose1d(64 -> 64, k, s, padding = s // 2) as a gather over its taps (network.py:67,86): weight (Ci, Co, k)"""
    g = torch.Generator(device=DEV).manual_seed(10 * k + s)
    N, NP, Lin = 300, 512, 6
    pad = s // 2
    Lo = (Lin - 1) * s - 2 * pad + k
    W = torch.randn(64, 64, k, generator=g, device=DEV) * 0.2         # (Ci, Co, k): ldw_m = k, ldw_c = Co * k, woff = tap
    x = _x(64, Lin, NP, N, g)
    c0, c1 = torch.rand(64, generator=g, device=DEV) + 0.5, torch.randn(64, generator=g, device=DEV) * 0.3
    segs = [dict(x=x, mul=1, off=pad - kk, div=s, woff=kk, c0=c0, c1=c1) for kk in range(k)]
    _run(64, segs, W, k, 64 * k, Lo, N, NP)


def test_x3_gru_projection_three_row_blocks():
    """M = 384 = three 128-row blocks of one launch (the FGRU input projection, network.py:48), plain prologue"""
    g = torch.Generator(device=DEV).manual_seed(9)
    N, NP, P = 500, 512, 4
    W = torch.randn(384, 128, generator=g, device=DEV) * 0.1
    seg = dict(x=_x(128, P, NP, N, g), mul=1, off=0, div=1, woff=0)
    _run(384, [seg], W, 128, 1, P, N, NP)


@pytest.mark.parametrize("name,N", [("tr_k3s1", 700), ("tr_k5s2", 520), ("first_tr", 300), ("dsc_k3s1", 650), ("dsc_k5s2", 777),
                                    ("dsc_k3s2", 300), ("last_tr", 420)])
def test_x3_fused_backward_kernels_vs_their_fp32_mfma_instances(name, N):
    """The fused backward kernels with both GEMMs on the bf16 MFMA through the three-term split (TRUNET_X3_BWD, the default
    since round 4) -- convt_bwd_x3_kernel<K, S> (ConvTranspose1d(64 -> 64) + BatchNorm) and pw_bwd_kernel<AK, SEC, KSPLIT, true>
    (Conv1d(k = 1) + BatchNorm: <64, ., KSPLIT> dsc_k3s1, <64> dsc_k5s2 / dsc_k3s2, <32, SEC> the decoder blocks, <32>
    first_tr, <16> last_tr) -- against their fp32-MFMA instances on the SAME recorded forward state and cotangent: the
    backward is linear given the state and both multiply the same fp32 numbers, so every parameter and input gradient of the
    block may differ by fp32 rounding only.  (Against float64 both are held by the block tests of test_network_gpu.py, which
    run on the split instances by default.)"""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_network_gpu import BLOCKS, BLOCK_SHAPES
    from oracle import weights as W
    from tinyrecurrentunet_amd import _lib as L, network as hn
    from tinyrecurrentunet_amd.engine import TRUNetEngine
    lib = L.lib()
    cls, args = BLOCKS[name]
    mod = W.fill_state_dict(getattr(hn, cls)(*args), seed=13).cuda().train()
    rng = np.random.default_rng(N)
    xs = [torch.tensor(rng.standard_normal(s) * 0.7, dtype=torch.float32).cuda() for s in BLOCK_SHAPES[name](N)]
    eng = TRUNetEngine(mod)
    prev = lib.trunet_gemm_x3_enable(-1)
    try:
        lib.trunet_gemm_x3_enable(0)
        y, ctx = eng.block_forward(mod._kind, getattr(mod, mod._seq), xs, True, record=True)
        gout = torch.tensor(rng.standard_normal(tuple(y.shape)), dtype=torch.float32).cuda()
        res = {}
        for mode in (2, 0):
            lib.trunet_gemm_x3_enable(mode)
            grads, gxs = eng.block_backward(ctx, gout)
            torch.cuda.synchronize()
            res[mode] = ({n: grads[p].clone() for n, p in mod.named_parameters() if p in grads}, [g.clone() for g in gxs])
    finally:
        lib.trunet_gemm_x3_enable(prev)
    worst = 0.0
    assert set(res[2][0]) == set(res[0][0]) and len(res[2][0]) >= 6
    for n, a in res[2][0].items():
        b = res[0][0][n]
        scale = b.norm().item()
        if scale < 1e-4 * b.numel() ** 0.5:        # conv bias in front of a BatchNorm: analytically zero, rounding noise
            assert a.norm().item() < 5e-2, (n, a.norm().item())
            continue
        e = (a - b).norm().item() / scale
        worst = max(worst, e)
        assert e < 2e-5, (n, e)
    for a, b in zip(res[2][1], res[0][1]):
        e = (a - b).norm().item() / b.norm().item()
        worst = max(worst, e)
        assert e < 2e-5, ("input gradient", e)
    assert not all(torch.equal(a, res[0][0][n]) for n, a in res[2][0].items()), "the toggle changed nothing"
    print("%s N=%d: split vs fp32-MFMA fused backward kernels, worst relative L2 %.2e" % (name, N, worst))


# ---------------------------------------------------------------- the opt-in path through the network (fp32 MFMA kind "bf16x3")
@pytest.fixture
def bf16x3():
    from tinyrecurrentunet_amd import _lib
    prev = _lib.set_fp32_mfma("bf16x3")
    yield
    _lib.set_fp32_mfma(prev)


@pytest.mark.parametrize("cin", [3, 4])
def test_bf16x3_network_forward_and_backward_match_reference_goldens(golden, cin, bf16x3):
    """With the split kernels on, the goldens of the reference composition hold unchanged: y_eval (layer-by-layer path)
    and y_train at 1e-4, every parameter gradient within the whole-network bounds of test_network_gpu._grad_close."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_network_gpu import _grad_close, _nets, _rel
    from tinyrecurrentunet_amd import _lib
    assert _lib.fp32_mfma() == "bf16x3"
    g = golden("trunet_cin%d" % cin)
    _, net = _nets(cin, seed=0)
    x = torch.tensor(g["x"]).cuda()
    net.eval()
    net.fold_eval = False
    with torch.no_grad():
        assert _rel(net(x), torch.tensor(g["y_eval"])) < 1e-4
    net.train()
    y = net(x)
    assert _rel(y, torch.tensor(g["y_train"])) < 1e-4
    (y * torch.tensor(g["cot"]).cuda()).sum().backward()
    for pn, p in net.named_parameters():
        if pn.startswith("TGRU"):
            continue
        _grad_close(p.grad, torch.tensor(g["g:" + pn]), pn)


def test_bf16x3_forward_vs_float64_and_vs_the_fp32_mfma_path(bf16x3):
    """Training-mode forward at a ragged frame count against the float64 oracle (1e-4 like the default path) and the
    parameter gradients of both fp32 paths against float64: the split path must not be further from the truth than the
    fp32-MFMA path (x 1.5 + the bounds' own noise floor)."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from oracle import network_ref as nr, weights as W
    from test_network_gpu import _nets, _rel
    from tinyrecurrentunet_amd import _lib
    N = 777
    _, net = _nets(4, seed=3)
    refd = W.fill_state_dict(nr.TRUNet(input_size=4), seed=3).double().train()
    x = torch.tensor(np.random.default_rng(N).standard_normal((N, 4, 257)) * 0.5, dtype=torch.float32)
    cot = torch.tensor(np.random.default_rng(N + 1).standard_normal((N, 8, 257)), dtype=torch.float32)
    yd = refd(x.double())
    (yd * cot.double()).sum().backward()
    pd = dict(refd.named_parameters())
    med = {}
    for kind in ("bf16x3", "fp32"):
        _lib.set_fp32_mfma(kind)
        net.zero_grad(set_to_none=True)
        net.train()
        y = net(x.cuda())
        assert _rel(y, yd) < 1e-4, (kind, _rel(y, yd))
        (y * cot.cuda()).sum().backward()
        es = []
        for pn, p in net.named_parameters():
            if pn.startswith("TGRU") or float(pd[pn].grad.abs().max()) < 1e-3:
                continue
            es.append(float((p.grad.double().cpu() - pd[pn].grad).norm() / pd[pn].grad.norm()))
        med[kind] = (float(np.median(es)), max(es))
    print("N = %d gradients vs float64: split median %.2e max %.2e; fp32-MFMA median %.2e max %.2e" % (
        (N,) + med["bf16x3"] + med["fp32"]))
    assert med["bf16x3"][0] < 1.5 * med["fp32"][0] + 5e-3 and med["bf16x3"][1] < 1.5 * med["fp32"][1] + 2e-2, med


def test_bf16x3_full_size_forward_agrees_with_the_fp32_mfma_path(bf16x3):
    """N = 32,064 (the benchmarked size; the fp32-MFMA forward at this size is pinned to the oracle at 1e-4 by
    test_configs_gpu.test_cfg2_*): every stored layer output of the split path within 1e-5 relative L2 and 3e-5 of max|z|
    of the fp32-MFMA path, the BatchNorm coefficients within 1e-6."""
    from tinyrecurrentunet_amd import _lib, network as hn
    from tinyrecurrentunet_amd.engine import TRUNetEngine
    torch.manual_seed(0)
    net = hn.TRUNet(input_size=4).cuda().train()
    N = 32064
    x = torch.randn(N, 4, 257, device="cuda") * 0.5
    res = {}
    for kind in ("fp32", "bf16x3"):
        _lib.set_fp32_mfma(kind)
        eng = TRUNetEngine(net)
        out, (acts, _, NP, w, _) = eng.forward(x, True, record=True)
        res[kind] = {k: (a.t[:, :, :N].clone(), None if a.bn is None else a.bn.scale.clone()) for k, a in acts.items()
                     if hasattr(a, "t")}
        del eng, acts, w
        torch.cuda.empty_cache()
    worst = 0.0
    for k, (b, sb) in res["fp32"].items():
        a, sa = res["bf16x3"][k]
        e = ((a - b).norm() / b.norm()).item()
        worst = max(worst, e)
        assert e < 1e-5 and ((a - b).abs().max() / b.abs().max()).item() < 3e-5, (k, e)
        if sb is not None:
            assert ((sa - sb).norm() / sb.norm()).item() < 1e-6, k
    print("full-size forward, split vs fp32-MFMA: worst layer relative L2 %.2e" % worst)
