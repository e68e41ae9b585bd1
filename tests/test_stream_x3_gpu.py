"""GPU: trunet_stream_fwd_x3 (stream_fwd_x3.hip: the single-launch eval forward with the encoder's pointwise layers on the bf16 MFMA
through the exact three-term split of the fp32 operands; the default kernel of export.FoldedTRUNet since round 4) against
trunet_stream_fwd (fp32 MFMA everywhere) and the float64 oracle: an fp32-grade result -- its error against float64 is the fp32
kernel's own."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _setup(cin, seed, tgru=False):
    from oracle import network_ref as nr, weights as W
    from tinyrecurrentunet_amd import export, network as hn
    ref = W.fill_state_dict(nr.TRUNet(input_size=cin), seed=seed)
    net = hn.TRUNet(input_size=cin)
    net.load_state_dict(ref.state_dict())
    net.cuda().eval()
    return ref, export.FoldedTRUNet.from_module(net, tgru=tgru)


@pytest.mark.parametrize("cin", [3, 4])
@pytest.mark.parametrize("N", [1, 255, 1024, 2500])
def test_split_kernel_vs_fp32_mfma_kernel_and_float64(N, cin):
    ref, f = _setup(cin, seed=2)
    assert f.x3 and f.blob_x3 is not None, "the split kernel is the default of FoldedTRUNet"
    g = torch.Generator().manual_seed(40 + N)
    x = torch.randn(N, cin, 257, generator=g)
    xg = x.cuda()
    y3 = f.use_x3(True)(xg)
    y0 = f.use_x3(False)(xg)
    assert not torch.equal(y3, y0), "the switch changed nothing"
    assert _rel(y3, y0) < 2e-6, _rel(y3, y0)
    if N <= 255:
        with torch.no_grad():
            yo = ref.double().eval()(x.double())
        e3, e0 = _rel(y3, yo), _rel(y0, yo)
        assert e3 < 1e-5 and e0 < 1e-5, (e3, e0)
        assert e3 < 2.0 * e0 + 1e-7, (e3, e0)
    # launch-to-launch determinism of the split kernel
    assert torch.equal(f.use_x3(True)(xg), y3)


def test_split_kernel_stateful_step_tracks_the_fp32_mfma_kernel():
    _, f = _setup(4, seed=5, tgru=True)
    h3, h0 = f.new_state(64), f.new_state(64)
    g = torch.Generator().manual_seed(7)
    for t in range(6):
        x = torch.randn(64, 4, 257, generator=g).cuda()
        y3 = f.use_x3(True).stream_step(x, h3)
        y0 = f.use_x3(False).stream_step(x, h0)
        assert _rel(y3, y0) < 5e-6, (t, _rel(y3, y0))
    assert _rel(h3, h0) < 1e-5


def test_split_image_is_an_exact_split_of_the_folded_weights():
    """export.x3_image: hi + mid + lo of every converted weight equals the folded fp32 weight bit for bit, every other section is
    copied verbatim, and the image passes the library's bounds check for the mask the library was built with."""
    import ctypes as C
    from tinyrecurrentunet_amd import _lib as L, export as E
    _, f = _setup(4, seed=3, tgru=True)
    blob = f.blob.cpu().numpy()
    mask = L.lib().trunet_stream_fwd_x3_mask()
    b3, o3 = E.x3_image(blob, f.offsets, mask)
    assert L.lib().trunet_stream_fwd_x3_check(o3.ctypes.data_as(C.POINTER(C.c_int32)), len(o3), len(b3), 4) == 0
    assert mask & 1
    for i in range(1, 6):
        M, K, tile, grp = E._X3_SECTIONS[i]
        W, b = E._unfrag_tiles(blob[f.offsets[i]:f.offsets[i + 1]], M, K)
        sec = b3[o3[i]:o3[i + 1]]
        KS, per = K // 16, (3 * (K // 16) + 4) * 256
        assert len(sec) == 4 * per
        lane = np.arange(64)
        for rt in range(4):
            t = sec[rt * per:(rt + 1) * per]
            planes = t[:3 * KS * 256].reshape(KS, 3, 64, 4)
            lo16 = (planes & 0xFFFF).astype(np.uint32) << 16
            hi16 = planes & np.uint32(0xFFFF0000)
            vals = np.stack([lo16, hi16], -1).reshape(KS, 3, 64, 8).view(np.float32)        # (ks, plane, lane, j)
            rec = vals[:, 0].astype(np.float64) + vals[:, 1] + vals[:, 2]
            rows = rt * 32 + (lane & 31)
            for ks in range(KS):
                k0 = 16 * ks + 8 * (lane >> 5)
                want = W[rows[:, None], k0[:, None] + np.arange(8)[None, :]]
                assert np.array_equal(rec[ks].astype(np.float32), want), (i, rt, ks)
    # a section outside the mask: verbatim
    assert np.array_equal(b3[o3[0]:o3[1]].view(np.float32), blob[f.offsets[0]:f.offsets[1]])
