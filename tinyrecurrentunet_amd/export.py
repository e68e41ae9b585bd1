"""Eval-mode export path (the role of ``/root/reference/onnx.py:14-44``: turn a checkpoint into
a deployable inference artefact, and of ``rt.py:13-27``: load it and run the 1-frame forward).

``fold(net)`` folds every BatchNorm (running statistics, eval semantics of network.py:31,39,51,65,72,...) into the conv
in front of it and lays the weights out for the single-launch forward kernel ``trunet_stream_fwd`` (stream_fwd.hip):
weight tiles in MFMA fragment order -- 32-row tiles (v_mfma_f32_32x32x2_f32: A fragments then 16 bias values per lane) for
the 128-channel encoder layers, 16-row tiles (v_mfma_f32_16x16x4_f32: A fragments then 4 bias values) for the GRU
projection, FGRU.conv and every decoder layer -- and per-tap matrices for the transposed convs.  The artefact is ONE flat fp32 tensor + 26 offsets; ``FoldedTRUNet.save`` / ``load`` store it with
``torch.save`` and read it back with ``weights_only=True``.  ``FoldedTRUNet.forward(x)`` is the whole network in one
kernel launch: (N, C_in, 257) -> (N, 8, 257), every frame independent (the reference's forward without TGRU, R4).

Round 3: ``fold(net, tgru=True)`` also exports the time-recurrent block (network.py:150, GRUBlock :45-58) for the stateful
causal stream of rt.py:20-27 / stream.py:83-109: ``FoldedTRUNet.stream_step(x, h)`` advances every (stream, frequency
position) sequence by ONE GRU time step inside the same single launch (between FGRU.conv and decoder.0, as drawn in
docs/net.jpg); the hidden state h (streams, 128, 16) lives in HBM, 8 KB per stream, read and written by the kernel."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L
from ._lib import check, ptr


def _bn_affine(bn):
    scale = bn.weight.detach().double().cpu() / torch.sqrt(bn.running_var.detach().double().cpu() + bn.eps)
    shift = bn.bias.detach().double().cpu() - bn.running_mean.detach().double().cpu() * scale
    return scale, shift


def _frag_tiles(Wm, bias):
    """Wm (M, K) float64, bias (M,): per 32-row tile [K/8 quads of A fragments][4 quads of bias], each quad [64 lanes][4].
    Lane l holds A[row l & 31][k = 2 kp + (l >> 5)] (v_mfma_f32_32x32x2_f32 A operand) and the bias of the 16 rows of
    its accumulator registers (row = (r & 3) + 8 (r >> 2) + 4 (l >> 5))."""
    M, K = Wm.shape
    nrt = (M + 31) // 32
    KP = (K + 15) // 16 * 8                      # k-pairs, multiple of 8 (the kernel's software pipeline)
    Wp = np.zeros((nrt * 32, 2 * KP))
    Wp[:M, :K] = Wm
    bp = np.zeros(nrt * 32)
    bp[:M] = bias
    lane = np.arange(64)
    out = []
    for rt in range(nrt):
        rows = rt * 32 + (lane & 31)                                          # (64,)
        kp = np.arange(KP)
        A = Wp[rows[None, :], 2 * kp[:, None] + (lane >> 5)[None, :]]        # (KP, 64)
        A = A.reshape(KP // 4, 4, 64).transpose(0, 2, 1)                      # (quad, lane, 4)
        r = np.arange(16)
        brow = rt * 32 + (r & 3)[:, None] + 8 * (r >> 2)[:, None] + 4 * (lane >> 5)[None, :]
        Bq = bp[brow].reshape(4, 4, 64).transpose(0, 2, 1)
        out.append(np.concatenate([A.reshape(-1), Bq.reshape(-1)]))
    return np.concatenate(out)


def _frag_tiles16(Wm, bias):
    """Wm (M, K) float64, bias (M,): per 16-row tile [K/16 quads of A fragments][1 quad of bias], each quad [64 lanes][4].
    Lane l holds A[row l & 15][k = 4 kq + (l >> 4)] (v_mfma_f32_16x16x4_f32 A operand) and the bias of the 4 rows of its
    accumulator registers (row = 4 (l >> 4) + r)."""
    M, K = Wm.shape
    nrt = (M + 15) // 16
    KQ = (K + 15) // 16 * 4
    Wp = np.zeros((nrt * 16, 4 * KQ))
    Wp[:M, :K] = Wm
    bp = np.zeros(nrt * 16)
    bp[:M] = bias
    lane = np.arange(64)
    out = []
    for rt in range(nrt):
        rows = rt * 16 + (lane & 15)
        kq = np.arange(KQ)
        A = Wp[rows[None, :], 4 * kq[:, None] + (lane >> 4)[None, :]]        # (KQ, 64)
        A = A.reshape(KQ // 4, 4, 64).transpose(0, 2, 1)                      # (quad, lane, 4)
        Bq = bp[rt * 16 + 4 * (lane >> 4)[:, None] + np.arange(4)[None, :]]   # (lane, 4)
        out.append(np.concatenate([A.reshape(-1), Bq.reshape(-1)]))
    return np.concatenate(out)


def _unfrag_tiles(sec, M, K):
    """Inverse of _frag_tiles: the section of a folded image -> (W (M, K), bias (M,)) float32."""
    nrt = (M + 31) // 32
    KP = (K + 15) // 16 * 8
    per = KP * 64 + 16 * 64
    sec = np.asarray(sec, dtype=np.float32)[:nrt * per].reshape(nrt, per)
    lane = np.arange(64)
    W = np.zeros((nrt * 32, 2 * KP), dtype=np.float32)
    b = np.zeros(nrt * 32, dtype=np.float32)
    r = np.arange(16)
    for rt in range(nrt):
        A = sec[rt, :KP * 64].reshape(KP // 4, 64, 4).transpose(0, 2, 1).reshape(KP, 64)          # (kp, lane)
        rows = rt * 32 + (lane & 31)
        W[rows[None, :], 2 * np.arange(KP)[:, None] + (lane >> 5)[None, :]] = A
        Bq = sec[rt, KP * 64:].reshape(4, 64, 4).transpose(0, 2, 1).reshape(16, 64)               # (r, lane)
        brow = rt * 32 + (r & 3)[:, None] + 8 * (r >> 2)[:, None] + 4 * (lane >> 5)[None, :]
        b[brow] = Bq
    return W[:M, :K], b[:M]


def _bf16_rne(x):
    """float32 array -> the nearest bf16 values (ties to even) as float32 (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    u = (u + (((u >> 16) & 1) + np.uint32(0x7FFF))) & np.uint32(0xFFFF0000)
    return u.view(np.float32)


def _split3(x):
    """x (float32) = hi + mid + lo, three bf16 terms rounded to nearest, the residues exact in fp32 (x3_common.hpp) -> three
    uint16 arrays (the bf16 bit patterns)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = _bf16_rne(x)
    r1 = x - hi
    mid = _bf16_rne(r1)
    lo = _bf16_rne(r1 - mid)
    return [(p.view(np.uint32) >> 16).astype(np.uint16) for p in (hi, mid, lo)]


def _frag_tiles_x3(Wm, bias):
    """Wm (M, K) float32, bias (M,): per 32-row tile [K/16 K-steps x 3 planes: one quad each][4 quads of bias (fp32 bits)], each
    quad [64 lanes][4 words].  Lane l holds the 8 bf16 values A[row l & 31][k = 16 ks + 8 (l >> 5) + 0..7] of a plane
    (v_mfma_f32_32x32x16_bf16 A operand; two values per word, the lower k in the lower half) -> uint32 array."""
    M, K = Wm.shape
    nrt = (M + 31) // 32
    KS = (K + 15) // 16
    Wp = np.zeros((nrt * 32, 16 * KS), dtype=np.float32)
    Wp[:M, :K] = Wm
    bp = np.zeros(nrt * 32, dtype=np.float32)
    bp[:M] = bias
    planes = _split3(Wp)
    lane = np.arange(64)
    r = np.arange(16)
    out = []
    for rt in range(nrt):
        rows = rt * 32 + (lane & 31)
        for ks in range(KS):
            k0 = 16 * ks + 8 * (lane >> 5)
            for pl in planes:
                v = pl[rows[:, None], k0[:, None] + np.arange(8)[None, :]].astype(np.uint32)      # (lane, 8)
                out.append((v[:, 0::2] | (v[:, 1::2] << 16)).reshape(-1))                         # (lane, 4 words)
        brow = rt * 32 + (r & 3)[:, None] + 8 * (r >> 2)[:, None] + 4 * (lane >> 5)[None, :]
        out.append(bp[brow].reshape(4, 4, 64).transpose(0, 2, 1).reshape(-1).view(np.uint32))
    return np.concatenate(out)


def _unfrag_tiles16(sec, M, K):
    """Inverse of _frag_tiles16 -> (W (M, K), bias (M,)) float32."""
    nrt = (M + 15) // 16
    KQ = (K + 15) // 16 * 4
    per = KQ * 64 + 256
    sec = np.asarray(sec, dtype=np.float32)[:nrt * per].reshape(nrt, per)
    lane = np.arange(64)
    W = np.zeros((nrt * 16, 4 * KQ), dtype=np.float32)
    b = np.zeros(nrt * 16, dtype=np.float32)
    for rt in range(nrt):
        A = sec[rt, :KQ * 64].reshape(KQ // 4, 64, 4).transpose(0, 2, 1).reshape(KQ, 64)          # (kq, lane)
        rows = rt * 16 + (lane & 15)
        W[rows[None, :], 4 * np.arange(KQ)[:, None] + (lane >> 4)[None, :]] = A
        b[rt * 16 + 4 * (lane >> 4)[:, None] + np.arange(4)[None, :]] = sec[rt, KQ * 64:].reshape(64, 4)
    return W[:M, :K], b[:M]


def _frag_tiles16_x3(Wm, bias):
    """Wm (M, K) float32, K a multiple of 32, bias (M,): per 16-row tile [K/32 K-steps x 3 planes: one quad each][1 quad of bias
    (fp32 bits)].  Lane l holds the 8 bf16 values A[row l & 15][k = 32 ks + 8 (l >> 4) + 0..7] of a plane
    (v_mfma_f32_16x16x32_bf16 A operand) -> uint32 array."""
    M, K = Wm.shape
    assert K % 32 == 0, K
    nrt = (M + 15) // 16
    KS = K // 32
    Wp = np.zeros((nrt * 16, K), dtype=np.float32)
    Wp[:M] = Wm
    bp = np.zeros(nrt * 16, dtype=np.float32)
    bp[:M] = bias
    planes = _split3(Wp)
    lane = np.arange(64)
    out = []
    for rt in range(nrt):
        rows = rt * 16 + (lane & 15)
        for ks in range(KS):
            k0 = 32 * ks + 8 * (lane >> 4)
            for pl in planes:
                v = pl[rows[:, None], k0[:, None] + np.arange(8)[None, :]].astype(np.uint32)
                out.append((v[:, 0::2] | (v[:, 1::2] << 16)).reshape(-1))
        out.append(np.ascontiguousarray(bp[rt * 16 + 4 * (lane >> 4)[:, None] + np.arange(4)[None, :]]).reshape(-1).view(np.uint32))
    return np.concatenate(out)


# matrix sections of the folded image: index -> (rows, K, 32- or 16-row tiles, layer group bit of stream_fwd_x3.hip's SFX_MASK)
_X3_SECTIONS = {1: (128, 64, 32, 1), 2: (128, 128, 32, 1), 3: (128, 128, 32, 1), 4: (128, 128, 32, 1), 5: (128, 128, 32, 1),
                11: (384, 128, 16, 16), 13: (64, 128, 16, 16), 14: (64, 64, 16, 16),
                15: (64, 192, 16, 2), 16: (64, 192, 16, 2), 17: (64, 192, 16, 2), 18: (64, 192, 16, 2), 19: (8, 128, 16, 16),
                20: (64, 192, 16, 16), 21: (64, 320, 16, 8), 22: (64, 192, 16, 4), 23: (64, 320, 16, 8), 24: (64, 192, 16, 4),
                26: (256, 192, 16, 16), 27: (128, 64, 16, 16), 28: (128, 128, 16, 16), 29: (64, 128, 16, 16)}


def x3_image(blob, offsets, mask):
    """The image `trunet_stream_fwd_x3` reads (round 4): the folded image with the matrix sections of the layer groups in `mask`
    (_X3_SECTIONS; the library reports the mask it was built for: trunet_stream_fwd_x3_mask) as three bf16 fragment planes -- an
    exact three-term split of the folded fp32 weights -- and everything else bit for bit -> (uint32 blob, offsets)."""
    blob = np.ascontiguousarray(np.asarray(blob, dtype=np.float32))
    offsets = np.asarray(offsets, dtype=np.int64)
    starts = sorted(set(int(offsets[i]) for i in range(N_OFFSETS) if i < 26 or offsets[i] > 0))
    bounds = {a: b for a, b in zip(starts, starts[1:] + [len(blob)])}
    sec, offs = [], []
    for i in range(N_OFFSETS):
        if i >= 26 and offsets[i] == 0:
            offs.append(0)
            continue
        a = int(offsets[i])
        raw = blob[a:bounds[a]]
        if i in _X3_SECTIONS and (_X3_SECTIONS[i][3] & mask):
            M, K, tile, _ = _X3_SECTIONS[i]
            if tile == 32:
                raw = _frag_tiles_x3(*_unfrag_tiles(raw, M, K))
            else:
                raw = _frag_tiles16_x3(*_unfrag_tiles16(raw, M, K))
        else:
            raw = raw.view(np.uint32)
        offs.append(sum(len(x) for x in sec))
        sec.append(raw)
    sec.append(np.zeros(64 * 256, dtype=np.uint32))     # fixed-size fragment requests over-read: they stay inside the image
    return np.concatenate(sec).astype(np.uint32), np.array(offs, dtype=np.int32)


# which kernel a FoldedTRUNet launches by default (FoldedTRUNet.use_x3 switches an instance)
STREAM_X3 = os.environ.get("TRUNET_STREAM_X3", "1") == "1"

N_OFFSETS = 30          # 26 sections of the stateless forward + 4 of the time-recurrent block (0 when not exported)


def fold(net, tgru=False):
    """TRUNet (network.py R1 layer sizes) -> (blob float32 ndarray, offsets int32[30], C_in).  Offsets 26..29 are the
    TGRU sections (0 = not exported): [W_ih | W_hh] rows of the r and z gates as sixteen 16-row tiles over K = 64 + 128
    (bias b_ih + b_hh), W_ih rows of the n gate (8 tiles, K = 64, bias b_in), W_hh rows of the n gate (8 tiles, K = 128,
    bias b_hn: it sits inside r * (.)), TGRU.conv + BatchNorm folded (4 tiles, K = 128)."""
    sec, offs = [], []

    def add(arr):
        arr = np.asarray(arr, dtype=np.float64).reshape(-1)
        pad = (-len(arr)) % 4
        offs.append(sum(len(s) for s in sec))
        sec.append(np.concatenate([arr, np.zeros(pad)]))

    def t(x):
        return x.detach().double().cpu().numpy()

    c0 = net.encoder[0].StandardConv1d[0]
    cin = c0.in_channels
    add(np.concatenate([t(c0.weight).reshape(-1), t(c0.bias)]))                       # o_first
    pws, dws = [], []
    for i in range(1, 6):
        seq = net.encoder[i].DepthwiseSeparableConv1d
        sc, sh = _bn_affine(seq[1])
        W = t(seq[0].weight)[:, :, 0] * sc.numpy()[:, None]
        pws.append(_frag_tiles(W, t(seq[0].bias) * sc.numpy() + sh.numpy()))
        sc, sh = _bn_affine(seq[4])
        Wd = t(seq[3].weight)[:, 0, :] * sc.numpy()[:, None]
        dws.append(np.concatenate([Wd.reshape(-1), t(seq[3].bias) * sc.numpy() + sh.numpy()]))
    for a in pws:
        add(a)                                                                         # o_pw[5]
    for a in dws:
        add(a)                                                                         # o_dw[5]
    g = net.FGRU.GRU
    Wih = np.concatenate([t(g.weight_ih_l0), t(g.weight_ih_l0_reverse)], 0)
    bih = np.concatenate([t(g.bias_ih_l0), t(g.bias_ih_l0_reverse)], 0)
    add(_frag_tiles16(Wih, bih))                                                       # o_gi
    # recurrence weights in the order the kernel's threads read them: [direction][24 quads][128 threads][4]; thread
    # t = 2 j + kh owns the K-half [32 kh, 32 kh + 32) of rows j (r), 64 + j (z), 128 + j (n): quad 8 g + i holds
    # W_hh[64 g + j][32 kh + 4 i .. + 3]; then b_hh of both directions
    whh = []
    for W in (t(g.weight_hh_l0), t(g.weight_hh_l0_reverse)):
        blk = np.zeros((24, 128, 4))
        for tt in range(128):
            j, kh = tt >> 1, tt & 1
            for gg in range(3):
                blk[8 * gg:8 * gg + 8, tt, :] = W[64 * gg + j, 32 * kh:32 * kh + 32].reshape(8, 4)
        whh.append(blk.reshape(-1))
    add(np.concatenate(whh + [t(g.bias_hh_l0), t(g.bias_hh_l0_reverse)]))                          # o_whh
    sc, sh = _bn_affine(net.FGRU.conv[1])
    fc = net.FGRU.conv[0]
    add(_frag_tiles16(t(fc.weight)[:, :, 0] * sc.numpy()[:, None], t(fc.bias) * sc.numpy() + sh.numpy()))   # o_fg
    dpw, cts, last = [], [], None
    for i in range(6):
        seq = net.decoder[i].FirstTrCNN if i == 0 else (net.decoder[i].TrCNN if i < 5 else net.decoder[i].LastTrCNN)
        sc, sh = _bn_affine(seq[1])
        dpw.append(_frag_tiles16(t(seq[0].weight)[:, :, 0] * sc.numpy()[:, None], t(seq[0].bias) * sc.numpy() + sh.numpy()))
        ct = seq[3]
        if i < 5:
            sc, sh = _bn_affine(seq[4])
            Wt = t(ct.weight) * sc.numpy()[None, :, None]                              # (Ci, Co, k)
            A = np.concatenate([Wt[:, :, k].T for k in range(Wt.shape[2])], 1)         # (Co, k*Ci): tap-major K axis
            cts.append(_frag_tiles16(A, t(ct.bias) * sc.numpy() + sh.numpy()))
        else:
            last = np.concatenate([t(ct.weight).reshape(-1), t(ct.bias)])              # linear output layer
    for a in dpw:
        add(a)                                                                         # o_dpw[6]
    for a in cts:
        add(a)                                                                         # o_ct[5]
    add(last)                                                                          # o_last
    assert len(offs) == 26
    if tgru:
        tg = net.TGRU.GRU
        if tg.bidirectional or tg.input_size != 64 or tg.hidden_size != 128:
            raise L.TrunetHipError("the exported time-recurrent block is the GRUBlock(64, 128, 64) of network.py:150")
        Wih, Whh, bih, bhh = t(tg.weight_ih_l0), t(tg.weight_hh_l0), t(tg.bias_ih_l0), t(tg.bias_hh_l0)
        H = 128
        add(_frag_tiles16(np.concatenate([Wih[:2 * H], Whh[:2 * H]], 1), bih[:2 * H] + bhh[:2 * H]))    # o_tg_rz
        add(_frag_tiles16(Wih[2 * H:], bih[2 * H:]))                                                    # o_tg_in
        add(_frag_tiles16(Whh[2 * H:], bhh[2 * H:]))                                                    # o_tg_hn
        sc, sh = _bn_affine(net.TGRU.conv[1])
        tc = net.TGRU.conv[0]
        add(_frag_tiles16(t(tc.weight)[:, :, 0] * sc.numpy()[:, None], t(tc.bias) * sc.numpy() + sh.numpy()))   # o_tg_conv
    else:
        offs.extend([0, 0, 0, 0])
    assert len(offs) == N_OFFSETS
    sec.append(np.zeros(64 * 256))         # the kernel requests fixed-size fragment blocks: over-reads stay inside the blob
    return np.concatenate(sec).astype(np.float32), np.array(offs, dtype=np.int32), cin


class FoldedTRUNet:
    """The exported inference artefact and its runner (one kernel launch per forward)."""

    def __init__(self, blob, offsets, cin, device=None):
        blob = torch.as_tensor(blob, dtype=torch.float32)
        offsets = np.ascontiguousarray(np.asarray(offsets, dtype=np.int32))
        if blob.dim() != 1 or offsets.shape != (N_OFFSETS,) or int(cin) not in (3, 4):
            raise L.TrunetHipError("not a folded TRU-Net image: blob %s, %s offsets, cin %r" % (
                tuple(blob.shape), offsets.shape, cin))
        # host-side twin of the entry point's bounds check: a truncated or foreign artefact must not reach the kernel
        rc = L.lib().trunet_stream_fwd_check(offsets.ctypes.data_as(C.POINTER(C.c_int32)), len(offsets), blob.numel(),
                                             int(cin))
        if rc != L.TRUNET_OK:
            raise L.TrunetHipError("folded TRU-Net image fails the section bounds check (truncated or foreign artefact)")
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.blob = blob.to(dev).contiguous()
        self.offsets = offsets
        self._offs = (C.c_int32 * len(self.offsets))(*[int(v) for v in self.offsets])
        self.cin = int(cin)
        self.has_tgru = bool(offsets[26] > 0)
        self._scratch = None
        # the same image with the encoder's pointwise weights as three bf16 fragment planes, for trunet_stream_fwd_x3
        # (STREAM_X3: which kernel forward() launches)
        self.x3 = STREAM_X3
        self.blob_x3 = self._offs_x3 = None
        if self.x3:
            self._make_x3()

    def _make_x3(self):
        b3, o3 = x3_image(self.blob.cpu().numpy(), self.offsets, L.lib().trunet_stream_fwd_x3_mask())
        rc = L.lib().trunet_stream_fwd_x3_check(o3.ctypes.data_as(C.POINTER(C.c_int32)), len(o3), len(b3), self.cin)
        if rc != L.TRUNET_OK:
            raise L.TrunetHipError("internal: the bf16-plane image fails its own bounds check")
        self.blob_x3 = torch.from_numpy(b3.view(np.int32).copy()).to(self.blob.device)
        self.offsets_x3 = o3
        self._offs_x3 = (C.c_int32 * len(o3))(*[int(v) for v in o3])

    def use_x3(self, on=True):
        """Select the kernel of forward() / stream_step(): trunet_stream_fwd_x3 (encoder pointwise layers on the bf16 MFMA through
        the three-term split) or trunet_stream_fwd (fp32 MFMA everywhere)."""
        self.x3 = bool(on)
        if self.x3 and self.blob_x3 is None:
            self._make_x3()
        return self

    @classmethod
    def from_module(cls, net, device=None, tgru=False):
        blob, offs, cin = fold(net, tgru=tgru)
        dev = device if device is not None else next(net.parameters()).device
        return cls(blob, offs, cin, dev)

    def save(self, path):
        torch.save({"format": "trunet-folded-v3", "blob": self.blob.cpu(), "offsets": torch.tensor(self.offsets),
                    "cin": self.cin}, path)

    @classmethod
    def load(cls, path, device=None):
        return cls.from_dict(torch.load(path, map_location="cpu", weights_only=True), device, what=path)

    @classmethod
    def from_dict(cls, d, device=None, what="dictionary"):
        """The saved dictionary -> runner.  "trunet-folded-v3" is what ``save`` writes; a "trunet-folded-v2" image (rounds 1-2:
        the 26 sections of the stateless forward, same layout) is a v3 image without the time-recurrent block -- its four
        TGRU offsets are 0 -- and is upgraded on load."""
        fmt = d.get("format") if isinstance(d, dict) else None
        if fmt not in ("trunet-folded-v3", "trunet-folded-v2"):
            raise L.TrunetHipError("%s is not a folded TRU-Net artefact (format %r)" % (what, fmt))
        offs = np.asarray(d["offsets"].numpy() if torch.is_tensor(d["offsets"]) else d["offsets"], dtype=np.int32)
        if fmt == "trunet-folded-v2":
            if offs.shape != (26,):
                raise L.TrunetHipError("%s: a v2 artefact has 26 section offsets, got %s" % (what, offs.shape))
            offs = np.concatenate([offs, np.zeros(N_OFFSETS - 26, dtype=np.int32)])
        return cls(d["blob"], offs, int(d["cin"]), device)

    def _run(self, x, h_in, h_out):
        if not x.is_cuda:
            raise L.TrunetHipError("tinyrecurrentunet_amd runs on MI355X only: got a %s tensor" % x.device)
        x = x.contiguous().float()
        if x.dim() != 3 or x.shape[1] != self.cin or x.shape[2] != 257:
            raise ValueError("expected (N, %d, 257) features, got %s" % (self.cin, tuple(x.shape)))
        N = x.shape[0]
        lib = L.lib()
        need = lib.trunet_stream_fwd_scratch_floats(lib.trunet_stream_fwd_grid(N))
        if self._scratch is None or self._scratch.numel() < need or self._scratch.device != x.device:
            self._scratch = torch.empty(need, device=x.device, dtype=torch.float32)
        y = torch.empty((N, 8, 257), device=x.device, dtype=torch.float32)
        if self.x3:
            check(lib.trunet_stream_fwd_x3(ptr(x), ptr(y), self.blob_x3.data_ptr(), self._offs_x3, len(self.offsets_x3),
                                           self.blob_x3.numel(), ptr(self._scratch), ptr(h_in), ptr(h_out), N, self.cin,
                                           L.stream()), "stream_fwd_x3")
        else:
            check(lib.trunet_stream_fwd(ptr(x), ptr(y), ptr(self.blob), self._offs, len(self.offsets), self.blob.numel(),
                                        ptr(self._scratch), ptr(h_in), ptr(h_out), N, self.cin, L.stream()), "stream_fwd")
        return y

    def forward(self, x):
        return self._run(x, None, None)

    __call__ = forward

    def new_state(self, streams, device=None):
        """h0 = 0 like nn.GRU: (streams, 128 hidden units, 16 frequency positions) fp32"""
        dev = device if device is not None else self.blob.device
        return torch.zeros((streams, 128, 16), device=dev, dtype=torch.float32)

    def stream_step(self, x, h):
        """One new STFT frame per stream, time-recurrent block included: x (streams, C_in, 257), h the state of
        ``new_state`` -- updated IN PLACE (every (stream, unit, position) element is read and written by the same thread
        of the same workgroup), so the state keeps its address and a captured hipGraph replays frame after frame."""
        if not self.has_tgru:
            raise L.TrunetHipError("this artefact was exported without the time-recurrent block: fold(net, tgru=True)")
        if h.shape != (x.shape[0], 128, 16) or h.dtype != torch.float32 or not h.is_cuda or not h.is_contiguous():
            raise ValueError("state must be the contiguous fp32 (streams, 128, 16) tensor of new_state()")
        return self._run(x, h, h)
