"""Host logic of the eval-mode exporter (tinyrecurrentunet_amd/export.py; role of /root/reference/onnx.py:14-44): BatchNorm
folding and the MFMA fragment layouts that stream_fwd.hip reads, checked by decoding the blob on the CPU (no GPU needed;
the kernel itself is checked against the y_eval goldens in tests/test_configs_gpu.py)."""
import numpy as np
import pytest
import torch

from tinyrecurrentunet_amd import _lib as L
from tinyrecurrentunet_amd import export
from tinyrecurrentunet_amd import network as hn


def _net():
    torch.manual_seed(3)
    net = hn.TRUNet(input_size=4).eval()
    with torch.no_grad():       # running statistics that are not the identity, so that folding is visible
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.uniform_(-0.5, 0.5)
                m.running_var.uniform_(0.5, 2.0)
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    return net


def _fold_conv_bn(conv_w, conv_b, bn):
    sc = bn.weight.detach().double() / torch.sqrt(bn.running_var.double() + bn.eps)
    sh = bn.bias.detach().double() - bn.running_mean.double() * sc
    return (conv_w.double() * sc[:, None]).numpy(), (conv_b.double() * sc + sh).numpy()


def _decode32(sec, M, K):
    """inverse of export._frag_tiles: per 32-row tile [K/8 quads of A][4 quads of bias], quad = [64 lanes][4]"""
    KP = (K + 15) // 16 * 8
    nq = KP // 4 + 4
    W, b = np.zeros((M, K)), np.zeros(M)
    for rt in range((M + 31) // 32):
        t = sec[rt * nq * 256:(rt + 1) * nq * 256].reshape(nq, 64, 4)
        for lane in range(64):
            row, h = rt * 32 + (lane & 31), lane >> 5
            for kp in range(KP):
                k = 2 * kp + h
                if row < M and k < K:
                    W[row, k] = t[kp // 4, lane, kp % 4]
            for r in range(16):
                brow = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h
                if brow < M:
                    b[brow] = t[KP // 4 + r // 4, lane, r % 4]
    return W, b


def _decode16(sec, M, K):
    """inverse of export._frag_tiles16: per 16-row tile [K/16 quads of A][1 quad of bias]"""
    KQ = (K + 15) // 16 * 4
    nq = KQ // 4 + 1
    W, b = np.zeros((M, K)), np.zeros(M)
    for rt in range((M + 15) // 16):
        t = sec[rt * nq * 256:(rt + 1) * nq * 256].reshape(nq, 64, 4)
        for lane in range(64):
            row, q = rt * 16 + (lane & 15), lane >> 4
            for kq in range(KQ):
                k = 4 * kq + q
                if row < M and k < K:
                    W[row, k] = t[kq // 4, lane, kq % 4]
            for r in range(4):
                brow = rt * 16 + 4 * q + r
                if brow < M:
                    b[brow] = t[KQ // 4, lane, r]
    return W, b


def test_fold_layout_and_batchnorm_folding():
    net = _net()
    blob, offs, cin = export.fold(net)
    assert cin == 4 and blob.dtype == np.float32 and offs.dtype == np.int32 and len(offs) == 30
    assert np.all(offs[26:] == 0)                                        # time-recurrent block not exported by default
    offs = offs[:26]
    assert np.all(offs % 4 == 0) and np.all(np.diff(offs) > 0)          # 16-byte aligned sections, in kernel order
    assert len(blob) >= offs[-1] + 8 * 8 * 5 + 8 + 64 * 256               # fixed-size requests never leave the blob
    o_first, o_pw, o_dw = offs[0], offs[1:6], offs[6:11]
    o_gi, o_whh, o_fg, o_dpw, o_ct, o_last = offs[11], offs[12], offs[13], offs[14:20], offs[20:25], offs[25]

    # encoder.2 pointwise (128 x 128): 32-row tiles
    seq = net.encoder[2].DepthwiseSeparableConv1d
    Wr, br = _fold_conv_bn(seq[0].weight.detach()[:, :, 0], seq[0].bias.detach(), seq[1])
    W, b = _decode32(blob[o_pw[1]:o_pw[2]].astype(np.float64), 128, 128)
    np.testing.assert_allclose(W, Wr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, br, rtol=1e-6, atol=1e-7)
    # its depthwise conv: [C][K] then bias
    Wd, bd = _fold_conv_bn(seq[3].weight.detach()[:, 0, :], seq[3].bias.detach(), seq[4])
    sec = blob[o_dw[1]:o_dw[1] + 128 * 5 + 128].astype(np.float64)
    np.testing.assert_allclose(sec[:640].reshape(128, 5), Wd, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(sec[640:], bd, rtol=1e-6, atol=1e-7)

    # GRU input projection (384 x 128, no BatchNorm) and FGRU.conv (64 x 128): 16-row tiles
    g = net.FGRU.GRU
    Wih = torch.cat([g.weight_ih_l0, g.weight_ih_l0_reverse]).detach().double().numpy()
    bih = torch.cat([g.bias_ih_l0, g.bias_ih_l0_reverse]).detach().double().numpy()
    W, b = _decode16(blob[o_gi:o_whh].astype(np.float64), 384, 128)
    np.testing.assert_allclose(W, Wih, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, bih, rtol=1e-6, atol=1e-7)
    fc, fbn = net.FGRU.conv[0], net.FGRU.conv[1]
    Wr, br = _fold_conv_bn(fc.weight.detach()[:, :, 0], fc.bias.detach(), fbn)
    W, b = _decode16(blob[o_fg:o_dpw[0]].astype(np.float64), 64, 128)
    np.testing.assert_allclose(W, Wr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, br, rtol=1e-6, atol=1e-7)

    # decoder.2: pointwise 192 -> 64 and the transposed conv (k = 3) as a tap-major (64 x 3*64) matrix
    seq = net.decoder[2].TrCNN
    Wr, br = _fold_conv_bn(seq[0].weight.detach()[:, :, 0], seq[0].bias.detach(), seq[1])
    W, b = _decode16(blob[o_dpw[2]:o_dpw[3]].astype(np.float64), 64, 192)
    np.testing.assert_allclose(W, Wr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, br, rtol=1e-6, atol=1e-7)
    ct, cbn = seq[3], seq[4]
    sc = cbn.weight.detach().double() / torch.sqrt(cbn.running_var.double() + cbn.eps)
    sh = cbn.bias.detach().double() - cbn.running_mean.double() * sc
    Wt = ct.weight.detach().double() * sc[None, :, None]                       # (Ci, Co, k)
    A = torch.cat([Wt[:, :, k].T for k in range(Wt.shape[2])], 1).numpy()
    W, b = _decode16(blob[o_ct[2]:o_ct[3]].astype(np.float64), 64, A.shape[1])
    np.testing.assert_allclose(W, A, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, (ct.bias.detach().double() * sc + sh).numpy(), rtol=1e-6, atol=1e-7)

    # decoder.5: pointwise 128 -> 8 (one padded 16-row tile), then the linear output layer verbatim
    seq = net.decoder[5].LastTrCNN
    Wr, br = _fold_conv_bn(seq[0].weight.detach()[:, :, 0], seq[0].bias.detach(), seq[1])
    W, b = _decode16(blob[o_dpw[5]:o_ct[0]].astype(np.float64), 8, 128)
    np.testing.assert_allclose(W, Wr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, br, rtol=1e-6, atol=1e-7)
    last = blob[o_last:o_last + 328].astype(np.float64)
    np.testing.assert_allclose(last[:320], seq[3].weight.detach().double().numpy().reshape(-1), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(last[320:], seq[3].bias.detach().double().numpy(), rtol=1e-6, atol=1e-7)


def test_recurrence_weights_are_in_thread_order():
    net = _net()
    blob, offs, _ = export.fold(net)
    g = net.FGRU.GRU
    sec = blob[offs[12]:offs[13]].astype(np.float64)
    for d, (W, bb) in enumerate(((g.weight_hh_l0, g.bias_hh_l0), (g.weight_hh_l0_reverse, g.bias_hh_l0_reverse))):
        blk = sec[d * 24 * 128 * 4:(d + 1) * 24 * 128 * 4].reshape(24, 128, 4)
        W = W.detach().double().numpy()
        for t in (0, 1, 37, 127):
            j, kh = t >> 1, t & 1
            for gg in range(3):
                np.testing.assert_allclose(blk[8 * gg:8 * gg + 8, t, :].reshape(-1), W[64 * gg + j, 32 * kh:32 * kh + 32],
                                           rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(sec[2 * 24 * 128 * 4 + d * 192:2 * 24 * 128 * 4 + (d + 1) * 192],
                                   bb.detach().double().numpy(), rtol=1e-6, atol=1e-7)


def test_tgru_sections_of_the_streaming_artefact():
    """fold(net, tgru=True): the time-recurrent block of network.py:150 as 16-row tiles -- r and z rows over the
    concatenated K = [x (64) | h (128)] with b_ih + b_hh, the n rows of W_ih and of W_hh apart (r multiplies W_hn h + b_hn
    only), TGRU.conv with its BatchNorm folded"""
    net = _net()
    blob, offs, _ = export.fold(net, tgru=True)
    assert len(offs) == 30 and np.all(offs[26:] > 0) and np.all(np.diff(offs) > 0) and np.all(offs % 4 == 0)
    o_rz, o_in, o_hn, o_cv = (int(v) for v in offs[26:])
    g = net.TGRU.GRU
    Wih, Whh = g.weight_ih_l0.detach().double().numpy(), g.weight_hh_l0.detach().double().numpy()
    bih, bhh = g.bias_ih_l0.detach().double().numpy(), g.bias_hh_l0.detach().double().numpy()
    W, b = _decode16(blob[o_rz:o_in].astype(np.float64), 256, 192)
    np.testing.assert_allclose(W, np.concatenate([Wih[:256], Whh[:256]], 1), rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, bih[:256] + bhh[:256], rtol=1e-6, atol=1e-7)
    W, b = _decode16(blob[o_in:o_hn].astype(np.float64), 128, 64)
    np.testing.assert_allclose(W, Wih[256:], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, bih[256:], rtol=1e-6, atol=1e-7)
    W, b = _decode16(blob[o_hn:o_cv].astype(np.float64), 128, 128)
    np.testing.assert_allclose(W, Whh[256:], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, bhh[256:], rtol=1e-6, atol=1e-7)
    tc, tbn = net.TGRU.conv[0], net.TGRU.conv[1]
    Wr, br = _fold_conv_bn(tc.weight.detach()[:, :, 0], tc.bias.detach(), tbn)
    W, b = _decode16(blob[o_cv:o_cv + 4 * 9 * 256].astype(np.float64), 64, 128)
    np.testing.assert_allclose(W, Wr, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(b, br, rtol=1e-6, atol=1e-7)


def test_section_bounds_check_refuses_truncated_and_foreign_images():
    """the offsets of an artefact are checked against the blob length (host side, no GPU): a truncated blob, an
    offset past the end, a misaligned or negative offset, a wrong C_in or a TGRU state request without the block are
    TRUNET_EINVAL, not a GPU fault"""
    import ctypes as C
    net = _net()
    lib = L.lib()

    def check(blob, offs, cin):
        arr = (C.c_int32 * len(offs))(*[int(v) for v in offs])
        return lib.trunet_stream_fwd_check(arr, len(offs), len(blob), cin)
    for tg in (False, True):
        blob, offs, cin = export.fold(net, tgru=tg)
        assert check(blob, offs, cin) == L.TRUNET_OK
        assert check(blob[:len(blob) - 20000], offs, cin) == L.TRUNET_EINVAL          # truncated
        assert check(blob[:int(offs[20])], offs, cin) == L.TRUNET_EINVAL
        bad = offs.copy(); bad[7] = len(blob) + 4
        assert check(blob, bad, cin) == L.TRUNET_EINVAL                               # section past the end
        bad = offs.copy(); bad[3] += 2
        assert check(blob, bad, cin) == L.TRUNET_EINVAL                               # not 16-byte aligned
        bad = offs.copy(); bad[0] = -4
        assert check(blob, bad, cin) == L.TRUNET_EINVAL
        assert check(blob, offs[:26], cin) == L.TRUNET_EINVAL                         # the 26-offset v2 layout
        assert check(blob, offs, 5) == L.TRUNET_EINVAL
    blob, offs, cin = export.fold(net)
    bad = offs.copy(); bad[28] = 64                                                   # partial TGRU export
    assert check(blob, bad, cin) == L.TRUNET_EINVAL
    with pytest.raises(L.TrunetHipError):
        export.FoldedTRUNet(blob[:1000], offs, cin, device="cpu")


def test_artefact_of_another_format_is_refused(tmp_path):
    p = tmp_path / "old.pt"
    torch.save({"format": "trunet-folded-v1", "blob": torch.zeros(8), "offsets": torch.zeros(26, dtype=torch.int32), "cin": 4}, p)
    with pytest.raises(L.TrunetHipError):
        export.FoldedTRUNet.load(str(p))
    # a v2 image must carry 26 offsets and still pass the section bounds check
    torch.save({"format": "trunet-folded-v2", "blob": torch.zeros(8), "offsets": torch.zeros(26, dtype=torch.int32), "cin": 4}, p)
    with pytest.raises(L.TrunetHipError):
        export.FoldedTRUNet.load(str(p), device="cpu")
    torch.save({"format": "trunet-folded-v2", "blob": torch.zeros(8), "offsets": torch.zeros(30, dtype=torch.int32), "cin": 4}, p)
    with pytest.raises(L.TrunetHipError):
        export.FoldedTRUNet.load(str(p), device="cpu")


def test_v2_artefact_is_upgraded_on_load(tmp_path):
    """artefacts written by rounds 1-2 ("trunet-folded-v2": the 26 sections of the stateless forward) are a v3
    image whose four TGRU offsets are 0: accepted on load, `has_tgru` False, same blob; `save` keeps writing v3"""
    net = _net()
    blob, offs, cin = export.fold(net)
    assert list(offs[26:]) == [0, 0, 0, 0]
    p = tmp_path / "v2.pt"
    torch.save({"format": "trunet-folded-v2", "blob": torch.tensor(blob), "offsets": torch.tensor(offs[:26]), "cin": cin}, p)
    run = export.FoldedTRUNet.load(str(p), device="cpu")          # host-side checks only: nothing is launched
    assert not run.has_tgru and run.cin == cin
    assert list(run.offsets) == list(offs)
    assert torch.equal(run.blob, torch.tensor(blob))
    q = tmp_path / "v3.pt"
    run.save(str(q))
    d = torch.load(str(q), weights_only=True)
    assert d["format"] == "trunet-folded-v3" and d["offsets"].numel() == export.N_OFFSETS
    again = export.FoldedTRUNet.load(str(q), device="cpu")
    assert torch.equal(again.blob, run.blob) and list(again.offsets) == list(run.offsets)
