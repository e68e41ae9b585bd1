"""Host-side schedule of the TRU-Net body on the HIP kernels (frames-last layout).

This is the MI355X-native replacement for what ATen does behind ``TRUNet.forward`` /
``loss.backward()`` in the reference (``/root/reference/network.py:122-171`` with repairs R1-R4
of SURVEY.md section 0.2).  Every tensor between layers is the RAW conv output (pre-BatchNorm);
BatchNorm+ReLU is applied by the consumer while it stages its operand, BatchNorm statistics
come from the producer's epilogue, and ``F.pad``/``torch.cat`` of the decoder are just row
offsets / a second operand segment -- nothing is copied.  Backward mirrors this: one dgrad launch
produces dy (ReLU-masked) plus the BatchNorm-backward sums, one wgrad launch per weight.

PyTorch is used for device memory only; all arithmetic is in libtrunet_hip.so.
"""
import torch

from . import _lib as L
import os

from ._lib import (DG_ACCUM, DG_MASK, DG_PREZERO, DG_STATS, DG_STORE, EPI_ACCUM, EPI_BIAS, EPI_MASK, EPI_PREZERO, EPI_STATS, PRO_BNBWD,
                   PRO_BNRELU, PRO_NONE, ConvtBwdArgs, GemmArgs, PwBwdArgs, WgradArgs, check, make_seg, ptr)

F_BINS = 257
FRAME_PAD = 256      # frames are padded to a multiple of 256 (widest conv_gemm tile)
BN_EPS = 1e-5
BN_MOM = 0.1


# Backward of the pointwise convs in front of a BatchNorm: one fused launch (trunet_pw_bwd) instead of
# trunet_conv_wgrad + trunet_conv_gemm; TRUNET_FUSED_PWBWD=0 keeps the separate launches (A/B measurements).
FUSED_PWBWD = os.environ.get("TRUNET_FUSED_PWBWD", "1") != "0"
# Depthwise backward without reading the conv's own output z (trunet_dwconv_bwd_rz recomputes it from the input rows, bit for
# bit what trunet_dwconv_fwd stored).  "0" / DW_RZ = False: read the stored z -- needed when the saved tensors are NOT this
# engine's own forward (tests that hand it the bf16 engine's rounded state).
DW_RZ = os.environ.get("TRUNET_DW_RZ", "1") != "0"

FUSED_THIN = os.environ.get("TRUNET_FUSED_THIN", "1") != "0"
FUSED_GRU_PROJ = os.environ.get("TRUNET_FUSED_GRU_PROJ", "1") != "0"

# Backward of the 64 -> 64 transposed convs (decoder.0 .. decoder.4): one fused launch (trunet_convt_bwd) instead of
# trunet_conv_wgrad + trunet_conv_gemm over the tap segments; TRUNET_FUSED_CONVT=0 keeps the separate launches.
FUSED_CONVT = os.environ.get("TRUNET_FUSED_CONVT", "1") != "0"

# TGRU time loop as a host loop of (GEMM, cell) launch pairs instead of the persistent kernels (A/B, other H)
TGRU_LOOP = os.environ.get("TRUNET_TGRU_LOOP", "0") == "1"

# bench.py sets this to a dict to time kernels with HIP events on the launch stream:
# PROFILE[kernel] = [(start_event, end_event, algorithmic_flops), ...]
PROFILE = None
PROFILE_LOG = None
PROFILE_BYTES = {}           # kernel name -> algorithmic HBM bytes of the profiled launches (set next to PROFILE)


class _Timed:
    def __init__(self, name, flops=0.0, tag="", nbytes=0.0):
        self.name, self.flops, self.tag, self.nbytes = name, flops, tag, nbytes

    def __enter__(self):
        if PROFILE is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record()
        return self

    def __exit__(self, *exc):
        if PROFILE is not None:
            self.b.record()
            PROFILE.setdefault(self.name, []).append((self.a, self.b, self.flops))
            if self.nbytes:          # algorithmic HBM bytes of the launch (kernels that are priced against the HBM roof)
                PROFILE_BYTES[self.name] = PROFILE_BYTES.get(self.name, 0.0) + self.nbytes
            if PROFILE_LOG is not None:
                PROFILE_LOG.append((self.name, self.tag, self.a, self.b, self.flops))
        return False


def _gemm_rs(M, segs):
    """Mirror of pick_rs() in gemm_conv.hip (which template instance a launch uses)."""
    nck = sum((s.nchan + 31) // 32 for s in segs)
    rs = 4 if M > 64 else (2 if M > 32 else 1)
    while rs > 1 and nck * rs * 4096 > 112 * 1024:
        rs >>= 1
    return rs


def _gemm_kernel_name(a):
    """Exact symbol (template arguments included) of the kernel trunet_conv_gemm launches for `a`, so that
    bench.py's per-kernel numbers can be matched against rocprofv3's kernel names."""
    import ctypes as C
    v = [C.c_int() for _ in range(6)]
    check(L.lib().trunet_conv_gemm_plan(a, *[C.byref(x) for x in v]), "conv_gemm_plan")
    rs, kc, nb, two, epl, nw = [x.value for x in v]
    if a.M <= 8 and not two:
        return "conv_smallm_kernel<%d>" % epl
    if nw < 0:              # the three-term bf16-split kernel (gemm_x3.hip)
        return "conv_gemm_x3_kernel<%d>" % rs
    return "conv_gemm_kernel<%d, %d, %s, %d, %d>" % (rs, kc, "true" if two else "false", epl, nw)


def _seg_positions(s, p0, P):
    """number of output positions p in [p0, p0+P) for which segment s is valid"""
    n = 0
    for p in range(p0, p0 + P):
        qn = p * s.pos_mul + s.pos_off
        if qn >= 0 and qn % s.pos_div == 0 and qn // s.pos_div < s.L:
            n += 1
    return n


def ceil_to(n, m):
    return (n + m - 1) // m * m


class BNState:
    """Per-layer BatchNorm affine (forward) and backward coefficients, all [C] device tensors."""

    def __init__(self, C, dev):
        z = lambda: torch.empty(C, device=dev, dtype=torch.float32)
        self.C = C
        self.scale, self.shift, self.mean, self.rstd = z(), z(), z(), z()
        self.ca, self.cb, self.cc = z(), z(), z()
        self.count = 0.0
        self.module = None


class Act:
    """A frames-last activation [C][L][NP] and how a consumer must read it."""

    def __init__(self, t, C, Ln, bn=None):
        self.t, self.C, self.L, self.bn = t, C, Ln, bn

    def seg(self, pos_off=0, woff=0, pos_mul=1, pos_div=1):
        if self.bn is None:
            return make_seg(self.t, self.C, self.L, pos_mul, pos_off, pos_div, woff, PRO_NONE)
        return make_seg(self.t, self.C, self.L, pos_mul, pos_off, pos_div, woff, PRO_BNRELU,
                        c0=self.bn.scale, c1=self.bn.shift)


class Workspace:
    """Name-keyed device buffers of one padded frame count.  Activations saved for backward live here too (they are the
    forward's own buffers), so a workspace that RECORDS for backward carries a generation counter: every recording
    forward bumps it, and ``backward`` refuses to run on buffers a later forward has overwritten."""

    def __init__(self, dev):
        self.dev = dev
        self.t = {}
        self.gen = 0
        # statistics partial buffers that a producer has written and no trunet_bn_finalize_* has consumed yet.  The finalize
        # kernels leave the rows they read ZERO, so a buffer that is not pending is all zero (flat(zero=True) at birth) and
        # its next producer can skip its zero-fill launch (TRUNET_EPI_PREZERO / TRUNET_DG_PREZERO: 46 launches per step);
        # after an exception between a producer and its finalize the name stays pending and the producer zero-fills again.
        self.pending = set()

    def take_clean(self, name):
        """True when partial buffer `name` is known to be all zero; either way it is marked as written"""
        clean = name not in self.pending
        self.pending.add(name)
        return clean

    def get(self, name, shape, zero=False, dtype=torch.float32):
        t = self.t.get(name)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = (torch.zeros if zero else torch.empty)(shape, device=self.dev, dtype=dtype)
            self.t[name] = t
        elif zero:
            t.zero_()
        return t

    def flat(self, name, numel, zero=False, dtype=torch.float32):
        """Grow-only 1-D scratch buffer (stream order makes sharing between layers safe); zero: zero-filled when it is
        (re)allocated, not per call."""
        t = self.t.get(name)
        if t is None or t.numel() < numel:
            t = (torch.zeros if zero else torch.empty)(max(numel, 1), device=self.dev, dtype=dtype)
            self.t[name] = t
            self.pending.discard(name)
        return t

    def zero_crop(self, t, q0, q1):
        """Positions [0, q0) and [q1, L) of gradient tensor t [C][L][...] are never written by any kernel (the cropped
        positions of network.py:96-97) and must read as zero: zero-filled once per tensor object, not once per step."""
        key = "crop0:%d" % id(t)
        if self.t.get(key) is not None and self.t[key][0] is t and self.t[key][1:] == (q0, q1):
            return
        if q0 > 0:
            t[:, :q0].zero_()
        if q1 < t.shape[1]:
            t[:, q1:].zero_()
        self.t[key] = (t, q0, q1)

    def bn(self, name, C):
        b = self.t.get("bn:" + name)
        if b is None:
            b = BNState(C, self.dev)
            self.t["bn:" + name] = b
        return b


class TRUNetEngine:
    # (kernel, stride) of the depthwise convs of encoder.1..5 and of the transposed convs of decoder.0..5
    ENC = [(3, 1), (5, 2), (3, 1), (5, 2), (3, 2)]
    DEC = [(3, 2), (5, 2), (3, 1), (5, 2), (3, 1), (5, 2)]

    def __init__(self, net):
        self.net = net
        self._ws = {}

    # ------------------------------------------------------------------ helpers
    def ws(self, NP, dev, record=False):
        """Workspace of (NP, device).  Forwards that record for backward and forwards that do not (no_grad / eval /
        streaming) get separate workspaces, so a validation forward between ``loss = net(x)`` and ``loss.backward()``
        cannot touch the saved activations; one size of each kind stays resident."""
        key = (NP, str(dev), bool(record))
        if key not in self._ws:
            self._ws = {k: v for k, v in self._ws.items() if k[2] != key[2]}
            self._ws[key] = Workspace(dev)
        return self._ws[key]

    @staticmethod
    def _check_gen(w, gen):
        if w.gen != gen:
            raise L.TrunetHipError(
                "backward() of a forward whose saved activations were overwritten: another gradient-recording forward of "
                "the same module ran before this backward (forward #%d, workspace is at #%d).  Call backward() before "
                "the next training forward (no_grad / eval forwards in between are fine)." % (gen, w.gen))

    def _gemm(self, w, *, N, NP, P, M, out, out_L, W, ldw_m, ldw_c, segs, p_begin=0, out_pos_off=0, m_out_off=0,
              w_m_off=0, epi=0, bias=None, zmask=None, e0=None, e1=None, e2=None, stats=None):
        a = GemmArgs()
        a.NP, a.N, a.P, a.p_begin = NP, N, P, p_begin
        a.M, a.m_out_off, a.out_L, a.out_pos_off = M, m_out_off, out_L, out_pos_off
        a.ldw_m, a.ldw_c, a.w_m_off = ldw_m, ldw_c, w_m_off
        a.nseg = len(segs)
        for i, s in enumerate(segs):
            a.seg[i] = s
        a.out, a.W = ptr(out), ptr(W)
        if bias is not None:
            epi |= EPI_BIAS
            a.bias = ptr(bias)
        if zmask is not None:
            epi |= EPI_MASK
            a.zmask, a.e0, a.e1, a.e2 = ptr(zmask), ptr(e0), ptr(e1), ptr(e2)
        nparts = 0
        if stats is not None:
            epi |= EPI_STATS
            nparts = L.lib().trunet_conv_gemm_nparts(M)
            part = w.flat("partials", nparts * stats * 2, zero=True)
            a.partials, a.M_stat = ptr(part), stats
            if w.take_clean("partials"):
                epi |= EPI_PREZERO
        a.epi = epi
        if PROFILE is not None:
            fl = 2.0 * N * M * sum(s.nchan * _seg_positions(s, p_begin, P) for s in segs)
            tag = "M%d K%s P%d%s%s" % (M, "+".join(str(s.nchan) for s in segs), P,
                                       " two" if any(s.mode == PRO_BNBWD for s in segs) else "",
                                       " mask" if zmask is not None else "")
            with _Timed(_gemm_kernel_name(a), fl, tag):
                check(L.lib().trunet_conv_gemm(a, L.stream()), "conv_gemm")
            return nparts
        check(L.lib().trunet_conv_gemm(a, L.stream()), "conv_gemm")
        return nparts

    def _gemm_args(self, *, N, NP, P, M, out, out_L, W, ldw_m, ldw_c, segs, bias=None, p_begin=0):
        """GemmArgs for a launch that is repeated with a changing segment offset (the TGRU time loop)."""
        a = GemmArgs()
        a.NP, a.N, a.P, a.p_begin = NP, N, P, p_begin
        a.M, a.m_out_off, a.out_L, a.out_pos_off = M, 0, out_L, 0
        a.ldw_m, a.ldw_c, a.w_m_off = ldw_m, ldw_c, 0
        a.nseg = len(segs)
        for i, sg in enumerate(segs):
            a.seg[i] = sg
        a.out, a.W = ptr(out), ptr(W)
        a.epi = 0
        if bias is not None:
            a.epi |= EPI_BIAS
            a.bias = ptr(bias)
        return a

    # ------------------------------------------------------------------ TGRU as a trained layer (use_tgru)
    def _tgru_block(self):
        """the GRUBlock(64, 128, 64, False) this engine runs over time: TRUNet.TGRU, or the module itself when the engine
        belongs to a stand-alone GRUBlock"""
        return getattr(self.net, "TGRU", self.net)

    def _tgru_seq_fwd(self, w, cur, N, NP, T, training, acts):
        """GRUBlock(64, 128, 64) of network.py:150 over TIME, between FGRU and the decoder (docs/net.jpg; SURVEY 8f
        rank 1): every (utterance, frequency position) is a sequence of T frames.  Sequence-major tensors [C][T][SP];
        the input projection, the block's pointwise conv and all weight gradients are single launches of the existing
        GEMM kernels, the recurrence itself is a host loop of (W_hh h GEMM, cell) pairs -- correct first; a persistent
        recurrent kernel is the next step."""
        lib, st = L.lib(), L.stream()
        blk = self._tgru_block()
        gru = blk.GRU
        Hh, Lf, C = gru.hidden_size, cur.L, cur.C
        if gru.bidirectional or gru.input_size != C or T <= 0 or N % T:
            raise L.TrunetHipError("use_tgru needs N = B*T frames (got N=%d, T=%d) and the GRUBlock(64,128,64) of "
                                   "network.py:150" % (N, T))
        B = N // T
        S = B * Lf
        SP = ceil_to(S, FRAME_PAD)
        xs = w.get("tg.xs", (C, T, SP))
        sc, sh = (cur.bn.scale, cur.bn.shift) if cur.bn is not None else (None, None)     # raw source: stand-alone block
        check(lib.trunet_to_seq_major(ptr(cur.t), ptr(xs), ptr(sc), ptr(sh), 1, C, Lf, T, B, NP, SP, st), "to_seq_major")
        gi = w.get("tg.gi", (3 * Hh, T, SP))
        persistent = Hh == 128 and not TGRU_LOOP
        b_in = gru.bias_ih_l0.data
        if persistent:      # the recurrent kernel wants b_hr, b_hz folded into the projection's bias (b_hn stays inside)
            b_in = w.get("tg.bias", (3 * Hh,))
            torch.add(gru.bias_ih_l0.data, gru.bias_hh_l0.data, out=b_in)
            b_in[2 * Hh:].copy_(gru.bias_ih_l0.data[2 * Hh:])
        self._gemm(w, N=S, NP=SP, P=T, M=3 * Hh, out=gi, out_L=T, W=gru.weight_ih_l0.data, ldw_m=C, ldw_c=1,
                   segs=[make_seg(xs, C, T)], bias=b_in)
        hs = w.get("tg.hs", (Hh, T + 1, SP))
        hs[:, 0].zero_()                                           # h_{-1} = 0 like nn.GRU
        gates = w.get("tg.gates", (4, Hh, T, SP)) if training else None
        if persistent:
            # persistent recurrence: one launch for all T steps (W_hh in registers, h through LDS)
            check(lib.trunet_tgru_rec_fwd(ptr(gi), ptr(gru.weight_hh_l0.data), ptr(gru.bias_hh_l0.data[2 * Hh:]),
                                          ptr(hs), ptr(gates), Hh, T, SP, st), "tgru_rec_fwd")
        else:
            gh = w.get("tg.gh", (3 * Hh, 1, SP))
            a = self._gemm_args(N=S, NP=SP, P=1, M=3 * Hh, out=gh, out_L=1, W=gru.weight_hh_l0.data, ldw_m=Hh,
                                ldw_c=1, segs=[make_seg(hs, Hh, T + 1)], bias=gru.bias_hh_l0.data)
            for t in range(T):
                a.seg[0].pos_off = t                               # h_{t-1} sits at position t
                check(lib.trunet_conv_gemm(a, st), "conv_gemm")
                check(lib.trunet_tgru_cell_fwd(ptr(gi), ptr(gh), ptr(hs), ptr(gates), Hh, T, t, SP, st),
                      "tgru_cell_fwd")
        # the block's Conv1d(128 -> 64, k=1) + BatchNorm over (B*16, 64, T), then back to frames-last
        conv, bn = blk.conv[0], blk.conv[1]
        zc = w.get("tg.zc", (conv.out_channels, T, SP))
        nparts = self._gemm(w, N=S, NP=SP, P=T, M=conv.out_channels, out=zc, out_L=T, W=conv.weight.data,
                            ldw_m=Hh, ldw_c=1, segs=[make_seg(hs, Hh, T + 1, pos_off=1)], bias=conv.bias.data,
                            stats=(conv.out_channels if training else None))
        stt = self._bn_fwd(w, "tgru", bn, conv.out_channels, S * T, nparts, training)
        zt = w.get("z:tgru", (conv.out_channels, Lf, NP))
        if NP > N:
            zt[:, :, N:].zero_()                                   # padding frames must stay finite (0 * NaN in the MFMAs)
        check(lib.trunet_from_seq_major(ptr(zc), ptr(zt), None, None, None, None, None, conv.out_channels, Lf, T, B, NP,
                                        SP, st), "from_seq_major")
        acts["tgru.ctx"] = dict(xs=xs, gi=gi, hs=hs, gates=gates, zc=zc, S=S, SP=SP, T=T, B=B)
        return Act(zt, conv.out_channels, Lf, stt)

    def _tgru_seq_bwd(self, w, acts, dy, bn, N, NP, grads):
        """Backward of _tgru_seq_fwd: dy [64][16][NP] is the (ReLU-masked) gradient of the block's BatchNorm output,
        bn that BatchNorm's state.  Returns the gradient w.r.t. FGRU's raw conv output (masked, with its statistics)."""
        lib, st = L.lib(), L.stream()
        c = acts["tgru.ctx"]
        xs, gi, hs, gates, zc, S, SP, T, B = (c[k] for k in ("xs", "gi", "hs", "gates", "zc", "S", "SP", "T", "B"))
        blk = self._tgru_block()
        gru, conv = blk.GRU, blk.conv[0]
        Hh, Cc = gru.hidden_size, conv.out_channels
        src = acts["fgru"]
        Lf, C = src.L, src.C
        dys = w.get("tg.dys", (Cc, T, SP))
        check(lib.trunet_to_seq_major(ptr(dy), ptr(dys), None, None, 0, Cc, Lf, T, B, NP, SP, st), "to_seq_major")
        # pointwise conv + BatchNorm backward (fused): weight/bias gradient and dL/dh_t at position t+1
        dhs = w.get("tg.dhs", (Hh, T + 1, SP))
        dhs[:, 0].zero_()
        self._pw_bwd(w, N=S, NP=SP, P=T, M=Cc, dz=dys, dz1=zc, dz_bn=bn, W=conv.weight, bias=conv.bias,
                     segs=[make_seg(hs, Hh, T + 1, pos_off=1)], outs=[dict(out=dhs)], grads=grads)
        # backward through time
        dgi = w.get("tg.dgi", (3 * Hh, T, SP))
        dgh = w.get("tg.dgh", (3 * Hh, T, SP))
        if Hh == 128 and not TGRU_LOOP:
            check(lib.trunet_tgru_rec_bwd(ptr(dhs), ptr(hs), ptr(gates), ptr(gru.weight_hh_l0.data), ptr(dgi), ptr(dgh),
                                          Hh, T, SP, S, st), "tgru_rec_bwd")
        else:
            cg = w.get("tg.carry", (Hh, 1, SP))
            a = self._gemm_args(N=S, NP=SP, P=1, M=Hh, out=cg, out_L=1, W=gru.weight_hh_l0.data, ldw_m=1, ldw_c=Hh,
                                segs=[make_seg(dgh, 3 * Hh, T)])
            carry = None
            for t in range(T - 1, -1, -1):
                check(lib.trunet_tgru_cell_bwd(ptr(dhs), ptr(carry), ptr(hs), ptr(gates), ptr(dgi), ptr(dgh), Hh, T, t,
                                               SP, S, st), "tgru_cell_bwd")
                if t > 0:
                    a.seg[0].pos_off = t                           # W_hh^T dgh_t -> gradient of h_{t-1}
                    check(lib.trunet_conv_gemm(a, st), "conv_gemm")
                    carry = cg
        # weight gradients over all time steps at once (384 rows of dz in three launches of 128)
        for g in range(3):
            self._wgrad(w, N=S, NP=SP, P=T, M=Hh, dz=dgh, dz_L=T, dz_bn=None, a_m_off=g * Hh, w_m_off=g * Hh,
                        W=gru.weight_hh_l0, ldw_m=Hh, ldw_c=1, segs=[make_seg(hs, Hh, T + 1)], grads=grads,
                        bias=gru.bias_hh_l0, b_off=g * Hh)
            self._wgrad(w, N=S, NP=SP, P=T, M=Hh, dz=dgi, dz_L=T, dz_bn=None, a_m_off=g * Hh, w_m_off=g * Hh,
                        W=gru.weight_ih_l0, ldw_m=C, ldw_c=1, segs=[make_seg(xs, C, T)], grads=grads,
                        bias=gru.bias_ih_l0, b_off=g * Hh)
        # data gradient of the input projection, back to frames-last through FGRU's BatchNorm+ReLU
        dxs = w.get("tg.dxs", (C, T, SP))
        self._gemm(w, N=S, NP=SP, P=T, M=C, out=dxs, out_L=T, W=gru.weight_ih_l0.data, ldw_m=1, ldw_c=C,
                   segs=[make_seg(dgi, 3 * Hh, T)])
        dyf = w.get("dy:fgru", (C, Lf, NP))
        if NP > N:
            dyf[:, :, N:].zero_()
        if src.bn is None:       # raw source (stand-alone block): plain gradient of the block input
            check(lib.trunet_from_seq_major(ptr(dxs), ptr(dyf), None, None, None, None, None, C, Lf, T, B, NP, SP, st),
                  "from_seq_major")
            return dyf, None, None
        nparts = lib.trunet_from_seq_major_nparts(Lf, T, B)
        part = w.flat("tg.partials", nparts * C * 2)
        check(lib.trunet_from_seq_major(ptr(dxs), ptr(dyf), ptr(src.t), ptr(src.bn.scale), ptr(src.bn.shift),
                                        ptr(src.bn.mean), ptr(part), C, Lf, T, B, NP, SP, st), "from_seq_major")
        self._bn_bwd(w, src.bn, nparts, grads, part_name="tg.partials")
        return dyf, src.t, src.bn

    def _bn_fwd(self, w, name, module, C, count, nparts, training):
        st = w.bn(name, C)
        st.module, st.count = module, float(count)
        lib = L.lib()
        if training:
            part = w.flat("partials", nparts * C * 2, zero=True)
            L.bump_mutation_epoch()     # running statistics are written through raw pointers
            rm = module.running_mean if module.track_running_stats else None
            rv = module.running_var if module.track_running_stats else None
            mom = BN_MOM if module.momentum is None else module.momentum
            nbt = module.num_batches_tracked.data_ptr() if module.track_running_stats else None
            check(lib.trunet_bn_finalize_fwd(ptr(part), nparts, C, float(count), ptr(module.weight.data),
                                             ptr(module.bias.data), module.eps, mom, ptr(rm), ptr(rv),
                                             ptr(st.scale), ptr(st.shift), ptr(st.mean), ptr(st.rstd), nbt, L.stream()),
                  "bn_finalize_fwd")
            w.pending.discard("partials")
        else:
            check(lib.trunet_bn_eval_affine(C, ptr(module.weight.data), ptr(module.bias.data),
                                            ptr(module.running_mean), ptr(module.running_var), module.eps,
                                            ptr(st.scale), ptr(st.shift), L.stream()), "bn_eval_affine")
        return st

    def _pw(self, w, name, srcs, conv, bn, N, NP, training, x1_left=0):
        """Conv1d(k=1) over one or two concatenated sources (+BN statistics).  srcs[0] may be
        shorter/longer than the output length (F.pad / crop by ``x1_left``, network.py:96-98)."""
        Ln = srcs[-1].L
        Co = conv.out_channels
        K = conv.in_channels
        out = w.get("z:" + name, (Co, Ln, NP))
        segs, off = [], 0
        for i, s in enumerate(srcs):
            segs.append(s.seg(pos_off=(-x1_left if (i == 0 and len(srcs) == 2) else 0), woff=off))
            off += s.C
        assert off == K
        nparts = self._gemm(w, N=N, NP=NP, P=Ln, M=Co, out=out, out_L=Ln, W=conv.weight.data, ldw_m=K, ldw_c=1,
                            segs=segs, bias=conv.bias.data, stats=(Co if (bn is not None and training) else None))
        st = self._bn_fwd(w, name, bn, Co, N * Ln, nparts, training) if bn is not None else None
        return Act(out, Co, Ln, st)

    def _convT(self, w, name, src, conv, bn, N, NP, training):
        k, s = conv.kernel_size[0], conv.stride[0]
        pad = conv.padding[0]
        Ci, Co = conv.in_channels, conv.out_channels
        Lo = (src.L - 1) * s - 2 * pad + k
        out = w.get("z:" + name, (Co, Lo, NP))
        segs = [src.seg(pos_off=pad - kk, woff=kk, pos_div=s) for kk in range(k)]
        nparts = self._gemm(w, N=N, NP=NP, P=Lo, M=Co, out=out, out_L=Lo, W=conv.weight.data, ldw_m=k,
                            ldw_c=Co * k, segs=segs, bias=conv.bias.data,
                            stats=(Co if (bn is not None and training) else None))
        st = self._bn_fwd(w, name, bn, Co, N * Lo, nparts, training) if bn is not None else None
        return Act(out, Co, Lo, st)

    def _dw(self, w, name, src, conv, bn, N, NP, training):
        k, s = conv.kernel_size[0], conv.stride[0]
        C = conv.out_channels
        Lo = (src.L + 2 * (k // 2) - k) // s + 1
        out = w.get("z:" + name, (C, Lo, NP))
        lib = L.lib()
        nparts = lib.trunet_dwconv_nparts(Lo)
        part = w.flat("partials_dw", nparts * C * 2)
        check(lib.trunet_dwconv_fwd(ptr(src.t), ptr(src.bn.scale), ptr(src.bn.shift), ptr(conv.weight.data),
                                    ptr(conv.bias.data), ptr(out), ptr(part), C, k, s, src.L, Lo, NP, N,
                                    L.stream()), "dwconv_fwd")
        st = w.bn(name, C)
        st.module, st.count = bn, float(N * Lo)
        if training:
            L.bump_mutation_epoch()     # running statistics are written through raw pointers
            rm = bn.running_mean if bn.track_running_stats else None
            rv = bn.running_var if bn.track_running_stats else None
            mom = BN_MOM if bn.momentum is None else bn.momentum
            nbt = bn.num_batches_tracked.data_ptr() if bn.track_running_stats else None
            check(lib.trunet_bn_finalize_fwd(ptr(part), nparts, C, float(N * Lo), ptr(bn.weight.data),
                                             ptr(bn.bias.data), bn.eps, mom, ptr(rm), ptr(rv), ptr(st.scale),
                                             ptr(st.shift), ptr(st.mean), ptr(st.rstd), nbt, L.stream()), "bn_finalize_fwd")
        else:
            check(lib.trunet_bn_eval_affine(C, ptr(bn.weight.data), ptr(bn.bias.data), ptr(bn.running_mean),
                                            ptr(bn.running_var), bn.eps, ptr(st.scale), ptr(st.shift), L.stream()),
                  "bn_eval_affine")
        return Act(out, C, Lo, st)


    # ------------------------------------------------------------------ stand-alone blocks (network.py:9-120)
    def _load(self, w, name, x):
        x = x.contiguous()
        N, C, Ln = x.shape
        NP = ceil_to(N, FRAME_PAD)
        t = w.get("in:" + name, (C, Ln, NP))
        check(L.lib().trunet_to_frames_last(ptr(x), ptr(t), N, C, Ln, NP, L.stream()), "to_frames_last")
        return Act(t, C, Ln), N, NP

    def _store(self, act, N, NP, dev, relu=True):
        out = torch.empty((N, act.C, act.L), device=dev, dtype=torch.float32)
        if act.bn is None:
            check(L.lib().trunet_from_frames_last(ptr(act.t), ptr(out), N, act.C, act.L, NP, L.stream()), "from_frames_last")
        else:
            check(L.lib().trunet_from_frames_last_affine(ptr(act.t), ptr(out), N, act.C, act.L, NP, ptr(act.bn.scale),
                                                         ptr(act.bn.shift), 1 if relu else 0, L.stream()),
                  "from_frames_last_affine")
        return out

    def block_forward(self, kind, seq, xs, training, record=False):
        """Forward of one reference block class (network.py:9-120) on (N, C, L) tensors; returns (y, ctx) with y the
        block's post-activation output like the reference and ctx what ``block_backward`` needs (record=True)."""
        dev = xs[0].device
        N = xs[0].shape[0]
        NP = ceil_to(N, FRAME_PAD)
        w = self.ws(NP, dev, record)
        if record:
            w.gen += 1
        lib = L.lib()
        acts = {}
        if kind == "std":
            a, N, NP = self._load(w, "b0", xs[0])
            conv = seq[0]
            k, s, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            if k != 5 or a.C not in (3, 4) or pad != s // 2:
                raise L.TrunetHipError("StandardConv1d kernel is built for C_in 3/4, k=5 (network.py:134)")
            Lo = (a.L + 2 * pad - k) // s + 1
            out = w.get("z:blk", (conv.out_channels, Lo, NP))
            check(lib.trunet_conv_first_fwd(ptr(a.t), ptr(conv.weight.data), ptr(conv.bias.data), ptr(out), a.C,
                                            conv.out_channels, k, s, a.L, Lo, NP, L.stream()), "conv_first_fwd")
            cur = Act(out, conv.out_channels, Lo)
            acts.update(x=a, out=cur)
        elif kind == "dsc":
            a, N, NP = self._load(w, "b0", xs[0])
            pw = self._pw(w, "blk.pw", [a], seq[0], seq[1], N, NP, training)
            cur = self._dw(w, "blk", pw, seq[3], seq[4], N, NP, training)
            acts.update(x=a, pw=pw, out=cur)
        elif kind in ("first_tr", "tr", "last_tr"):
            a, N, NP = self._load(w, "b0", xs[0])
            srcs, left, b = [a], 0, None
            if kind != "first_tr":
                b, _, _ = self._load(w, "b1", xs[1])
                left = (b.L - a.L) // 2
                srcs = [a, b]
            pw = self._pw(w, "blk.pw", srcs, seq[0], seq[1], N, NP, training, x1_left=left)
            cur = self._convT(w, "blk", pw, seq[3], seq[4] if kind != "last_tr" else None, N, NP, training)
            acts.update(x=a, x2=b, left=left, pw=pw, out=cur)
        else:
            raise ValueError(kind)
        return self._store(cur, N, NP, dev), (kind, seq, acts, N, NP, w, w.gen)

    def gru_block_forward(self, blk, x, training, record=False):
        """GRUBlock.forward (network.py:54-58): x (S, L, C_in) -> (S, C_out, L).  bidirectional H=64: the FGRU kernels
        (recurrence over the L positions of every frame); unidirectional H=128 (network.py:150's TGRU class): the
        time-recurrent kernels with every row of x as one sequence."""
        gru = blk.GRU
        dev = x.device
        S, Lx, C = x.shape
        if gru.bidirectional and gru.hidden_size == 64:
            NP = ceil_to(S, FRAME_PAD)
            w = self.ws(NP, dev, record)
            if record:
                w.gen += 1
            a, N, NP = self._load(w, "b0", x.transpose(1, 2))
            hout = self._gru(w, a, gru, N, NP, training)
            cur = self._pw(w, "blk.fgru", [hout], blk.conv[0], blk.conv[1], N, NP, training)
            acts = dict(x=a, hout=hout, out=cur)
            return self._store(cur, N, NP, dev), ("gru_bi", blk, acts, N, NP, w, w.gen)
        if not gru.bidirectional:
            # frames-last with ONE position per frame: frame index s*T + t, so (s, position 0) is a sequence over t
            N, T = S * Lx, Lx
            NP = ceil_to(N, FRAME_PAD)
            w = self.ws(NP, dev, record)
            if record:
                w.gen += 1
            a, N, NP = self._load(w, "b0", x.reshape(N, C, 1))
            acts = {"fgru": a}
            cur = self._tgru_seq_fwd(w, a, N, NP, T, training, acts)
            acts["out"] = cur
            y = self._store(cur, N, NP, dev)                           # (S*T, C_out, 1)
            return y.reshape(S, T, cur.C).transpose(1, 2).contiguous(), ("gru_uni", blk, acts, N, NP, w, w.gen)
        raise L.TrunetHipError("the HIP GRU kernels are built for GRUBlock(128, 64, 64, True) (network.py:149) and "
                               "GRUBlock(64, 128, 64, False) (network.py:150)")

    def _entry(self, w, out, gout, N, NP, grads, relu=True):
        """Cotangent of a block's post-activation output (N, C, L) -> upstream state (dy, z, bn) of the fused schedule:
        dy = gradient at the BatchNorm output with the ReLU mask applied, plus the BatchNorm-backward reduction."""
        lib, st = L.lib(), L.stream()
        gout = gout.contiguous().float()
        dyt = w.get("dy:blk", (out.C, out.L, NP))
        check(lib.trunet_to_frames_last(ptr(gout), ptr(dyt), N, out.C, out.L, NP, st), "to_frames_last")
        if out.bn is None:
            if relu:
                check(lib.trunet_relu_bwd_stats(ptr(dyt), ptr(out.t), None, None, None, None, out.C, out.L, NP, N, st),
                      "relu_bwd_stats")
            return dyt, out.t, None
        nparts = lib.trunet_relu_bwd_stats_nparts()
        part = w.flat("entry_partials", nparts * out.C * 2)
        check(lib.trunet_relu_bwd_stats(ptr(dyt), ptr(out.t), ptr(out.bn.scale), ptr(out.bn.shift), ptr(out.bn.mean),
                                        ptr(part), out.C, out.L, NP, N, st), "relu_bwd_stats")
        self._bn_bwd(w, out.bn, nparts, grads, part_name="entry_partials")
        return dyt, out.t, out.bn

    def _unload(self, t, N, C, Ln, NP):
        out = torch.empty((N, C, Ln), device=t.device, dtype=torch.float32)
        check(L.lib().trunet_from_frames_last(ptr(t), ptr(out), N, C, Ln, NP, L.stream()), "from_frames_last")
        return out

    def block_backward(self, ctx, gout):
        """Backward of block_forward / gru_block_forward: gout = cotangent of the block output.  Returns
        ({parameter: gradient}, [gradient of each block input])."""
        kind, seq, acts, N, NP, w, gen = ctx
        self._check_gen(w, gen)
        grads = {}
        self._wg_begin(w)
        out = acts["out"]
        if kind == "gru_uni":
            g = gout.transpose(1, 2).reshape(N, out.C, 1)
            up = self._entry(w, out, g, N, NP, grads)
            dyx, _, _ = self._tgru_seq_bwd(w, acts, up[0], up[2], N, NP, grads)
            x = acts["fgru"]
            gx = self._unload(dyx, N, x.C, 1, NP)
            T = acts["tgru.ctx"]["T"]
            gxs = [gx.reshape(N // T, T, x.C)]
        elif kind == "gru_bi":
            up = self._entry(w, out, gout, N, NP, grads)
            x = acts["x"]
            gxt = w.get("gx0", (x.C, x.L, NP))
            self._bwd_fgru(w, N, NP, seq, up, acts["hout"], x, None, gxt, grads)
            gxs = [self._unload(gxt, N, x.C, x.L, NP).transpose(1, 2)]
        elif kind == "std":
            conv = seq[0]
            x = acts["x"]
            dy, _, _ = self._entry(w, out, gout, N, NP, grads, relu=True)
            self._bwd_first(w, N, NP, conv, x, dy, out.L, grads)
            k, s_, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            gxt = w.get("gx0", (x.C, x.L, NP))       # (the full network never needs it: features carry no gradient)
            self._gemm(w, N=N, NP=NP, P=x.L, M=x.C, out=gxt, out_L=x.L, W=conv.weight.data, ldw_m=k, ldw_c=x.C * k,
                       segs=[make_seg(dy, out.C, out.L, pos_off=pad - kk, pos_div=s_, woff=kk) for kk in range(k)])
            gxs = [self._unload(gxt, N, x.C, x.L, NP)]
        elif kind == "dsc":
            up = self._entry(w, out, gout, N, NP, grads)
            x = acts["x"]
            gxt = w.get("gx0", (x.C, x.L, NP))
            self._bwd_dsc(w, N, NP, seq, acts["pw"], out, up, x, None, False, gxt, "dy:blk.pw", grads)
            gxs = [self._unload(gxt, N, x.C, x.L, NP)]
        else:
            up = self._entry(w, out, gout, N, NP, grads, relu=(kind != "last_tr"))
            x, x2 = acts["x"], acts["x2"]
            gx1 = w.get("gx0", (x.C, x.L, NP))
            gx2 = w.get("gx1", (x2.C, x2.L, NP)) if x2 is not None else None
            self._bwd_tr(w, N, NP, seq[3], seq[0], acts["pw"], out.L, up, x, None, x2, acts["left"], gx1, gx2,
                         "dy:blk.pw", grads)
            gxs = [self._unload(gx1, N, x.C, x.L, NP)]
            if x2 is not None:
                gxs.append(self._unload(gx2, N, x2.C, x2.L, NP))
        self._wg_finish(grads)
        return grads, gxs

    def _gru(self, w, cur, gru, N, NP, training):
        """input projection (both directions, M = 6H) + recurrence; returns Act(hout [2H][L][NP])"""
        lib = L.lib()
        Hh = gru.hidden_size
        wih = w.get("wih", (6 * Hh, gru.input_size))
        bih = w.get("bih", (6 * Hh,))
        torch.cat((gru.weight_ih_l0.data, gru.weight_ih_l0_reverse.data), 0, out=wih)
        torch.cat((gru.bias_ih_l0.data, gru.bias_ih_l0_reverse.data), 0, out=bih)
        Lg = cur.L
        gi = w.get("gi", (6 * Hh, Lg, NP))
        self._gemm(w, N=N, NP=NP, P=Lg, M=6 * Hh, out=gi, out_L=Lg, W=wih, ldw_m=gru.input_size, ldw_c=1,
                   segs=[cur.seg()], bias=bih)
        hout = w.get("hout", (2 * Hh, Lg, NP))
        gates = w.get("gates", (2, 4, Hh, Lg, NP)) if training else None
        check(lib.trunet_gru_fwd(ptr(gi), ptr(gru.weight_hh_l0.data), ptr(gru.bias_hh_l0.data),
                                 ptr(gru.weight_hh_l0_reverse.data), ptr(gru.bias_hh_l0_reverse.data), ptr(hout),
                                 ptr(gates), Hh, Lg, NP, L.stream()), "gru_fwd")
        return Act(hout, 2 * Hh, Lg)

    # ------------------------------------------------------------------ forward
    def _tgru_step(self, w, cur, state, N, NP):
        """One time step of the TGRU block (network.py:150, GRUBlock :45-58) per stream: every (stream, frequency
        position) pair is a sequence whose hidden state h [128][16][NP] is carried in ``state`` between calls (the
        paper-style use of the block; the reference constructs it and never calls it, D6)."""
        lib = L.lib()
        blk = self.net.TGRU
        gru = blk.GRU
        Hh, Lg = gru.hidden_size, cur.L
        if gru.bidirectional or gru.input_size != cur.C:
            raise L.TrunetHipError("TGRU streaming expects the unidirectional GRUBlock(64, 128, 64) of network.py:150")
        if state.h is not None and state.n != N:
            raise L.TrunetHipError("stream state belongs to %d streams, got %d" % (state.n, N))
        if state.h is None or tuple(state.h.shape) != (Hh, Lg, NP):
            state.h = torch.zeros((Hh, Lg, NP), device=cur.t.device, dtype=torch.float32)     # h0 = 0 like nn.GRU
            state.n = N
            if getattr(state, "layout", None) is None:
                state.layout = "frames_last"
        gi = w.get("tgru.gi", (3 * Hh, Lg, NP))
        gh = w.get("tgru.gh", (3 * Hh, Lg, NP))
        self._gemm(w, N=N, NP=NP, P=Lg, M=3 * Hh, out=gi, out_L=Lg, W=gru.weight_ih_l0.data, ldw_m=gru.input_size,
                   ldw_c=1, segs=[cur.seg()], bias=gru.bias_ih_l0.data)
        self._gemm(w, N=N, NP=NP, P=Lg, M=3 * Hh, out=gh, out_L=Lg, W=gru.weight_hh_l0.data, ldw_m=Hh, ldw_c=1,
                   segs=[Act(state.h, Hh, Lg).seg()], bias=gru.bias_hh_l0.data)
        # elementwise and in place (every element reads its own h): the state keeps its address, so a captured
        # hipGraph of one step can be replayed frame after frame
        check(lib.trunet_gru_cell(ptr(gi), ptr(gh), ptr(state.h), ptr(state.h), Hh, Lg, NP, L.stream()), "gru_cell")
        state.steps += 1
        return self._pw(w, "tgru", [Act(state.h, Hh, Lg)], blk.conv[0], blk.conv[1], N, NP, False)

    def forward(self, x, training, tgru_state=None, tgru_T=None, record=False):
        net = self.net
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == F_BINS
        x = x.contiguous()
        N, Cin = x.shape[0], x.shape[1]
        NP = ceil_to(N, FRAME_PAD)
        w = self.ws(NP, x.device, record)
        if record:
            w.gen += 1
        lib = L.lib()
        st = L.stream()
        acts = {}

        xt = w.get("x", (Cin, F_BINS, NP))
        check(lib.trunet_to_frames_last(ptr(x), ptr(xt), N, Cin, F_BINS, NP, st), "to_frames_last")
        c0 = net.encoder[0].StandardConv1d[0]
        assert c0.in_channels == Cin
        a0t = w.get("z:enc0", (64, 128, NP))
        check(lib.trunet_conv_first_fwd(ptr(xt), ptr(c0.weight.data), ptr(c0.bias.data), ptr(a0t), Cin, 64,
                                        c0.kernel_size[0], c0.stride[0], F_BINS, 128, NP, st), "conv_first_fwd")
        acts["x"] = Act(xt, Cin, F_BINS)
        cur = acts["enc0"] = Act(a0t, 64, 128)
        for i in range(1, 6):
            seq = net.encoder[i].DepthwiseSeparableConv1d
            cur = acts["enc%d.pw" % i] = self._pw(w, "enc%d.pw" % i, [cur], seq[0], seq[1], N, NP, training)
            cur = acts["enc%d" % i] = self._dw(w, "enc%d" % i, cur, seq[3], seq[4], N, NP, training)

        # FGRU: input projection (both directions, M = 384), recurrence, pw conv + BN
        gru = net.FGRU.GRU
        acts["hout"] = self._gru(w, cur, gru, N, NP, training)
        cur = acts["fgru"] = self._pw(w, "fgru", [acts["hout"]], net.FGRU.conv[0], net.FGRU.conv[1], N, NP, training)

        if tgru_state is not None:
            cur = acts["tgru"] = self._tgru_step(w, cur, tgru_state, N, NP)
        elif tgru_T is not None:
            cur = acts["tgru"] = self._tgru_seq_fwd(w, cur, N, NP, tgru_T, training, acts)

        seq = net.decoder[0].FirstTrCNN
        cur = acts["dec0.pw"] = self._pw(w, "dec0.pw", [cur], seq[0], seq[1], N, NP, training)
        cur = acts["dec0"] = self._convT(w, "dec0", cur, seq[3], seq[4], N, NP, training)
        for i in range(1, 6):
            skip = acts["enc%d" % (5 - i)]
            seq = net.decoder[i].TrCNN if i < 5 else net.decoder[i].LastTrCNN
            left = (skip.L - cur.L) // 2                       # F.pad left amount (negative = crop)
            cur = acts["dec%d.pw" % i] = self._pw(w, "dec%d.pw" % i, [cur, skip], seq[0], seq[1], N, NP, training,
                                                  x1_left=left)
            cur = acts["dec%d" % i] = self._convT(w, "dec%d" % i, cur, seq[3], seq[4] if i < 5 else None, N, NP,
                                                  training)
        out = torch.empty((N, cur.C, cur.L), device=x.device, dtype=torch.float32)
        check(lib.trunet_from_frames_last(ptr(cur.t), ptr(out), N, cur.C, cur.L, NP, st), "from_frames_last")
        return out, (acts, N, NP, w, w.gen)

    # ------------------------------------------------------------------ backward
    def _bn_bwd(self, w, st, nparts, grads, part_name="partials"):
        m = st.module
        part = w.flat(part_name, nparts * st.C * 2, zero=True)
        # dgamma / dbeta go straight into the parameters' slots of partial image 0 (the other images hold zeros there)
        check(L.lib().trunet_bn_finalize_bwd(ptr(part), nparts, st.C, st.count, ptr(m.weight.data), ptr(st.mean),
                                             ptr(st.rstd), self._wg_slot(m.weight), self._wg_slot(m.bias), ptr(st.ca),
                                             ptr(st.cb), ptr(st.cc), L.stream()), "bn_finalize_bwd")
        w.pending.discard(part_name)

    # ---- gradient partial images: every parameter owns a slice of ONE buffer [nparts][total] (slices start at multiples
    # of 64 elements).  The conv / GRU weight-gradient launches write their per-workgroup partial images straight into
    # it (image stride = total); BatchNorm and depthwise gradients, which have their own small reductions, are written
    # into image 0 only (the buffer is zero-filled when allocated and nothing else ever touches those slices or the
    # gaps).  A single trunet_reduce_partials at the end of backward sums all images into ONE flat gradient tensor: every
    # p.grad is a view of it, and the all-reduce / optimizer work on it in place (_lib.register_flat_grad).
    def _wg_begin(self, w):
        if getattr(self, "_wg_layout", None) is None:
            off, lay = 0, {}
            for p in self.net._active_params() if hasattr(self.net, "_active_params") else self.net.parameters():
                lay[p] = off
                off += (p.numel() + 63) // 64 * 64
            self._wg_layout, self._wg_total = lay, off
        self._wg_base = w.flat("wg_partials", L.lib().trunet_conv_wgrad_nparts() * self._wg_total, zero=True)
        self._wg_touched = {}

    def _wg_slot(self, p):
        """device pointer of parameter p's slice in partial image 0"""
        self._wg_touched[id(p)] = p
        return self._wg_base.data_ptr() + 4 * self._wg_layout[p]

    def _wg_finish(self, grads):
        lib = L.lib()
        flat = torch.empty(self._wg_total, device=self._wg_base.device, dtype=torch.float32)   # fresh: autograd keeps it
        check(lib.trunet_reduce_partials(ptr(flat), ptr(self._wg_base), lib.trunet_conv_wgrad_nparts(), self._wg_total,
                                         0, L.stream()), "reduce")
        for p in self._wg_touched.values():
            o = self._wg_layout[p]
            grads[p] = flat[o:o + p.numel()].view_as(p)
        if len(self._wg_touched) == len(self._wg_layout):
            L.register_flat_grad(flat, {id(p): o for p, o in self._wg_layout.items()}, self._wg_total)

    def _wgrad(self, w, *, N, NP, P, M, dz, dz_L, dz_bn, W, ldw_m, ldw_c, segs, grads, bias=None, a_pos_off=0,
               a_m_off=0, w_m_off=0, b_off=0, dz1=None):
        lib = L.lib()
        a = WgradArgs()
        a.NP, a.N, a.P, a.p_begin = NP, N, P, 0
        a.M, a.a_L, a.a_pos_off, a.a_m_off = M, dz_L, a_pos_off, a_m_off
        a.ldw_m, a.ldw_c, a.w_m_off = ldw_m, ldw_c, w_m_off
        a.nseg = len(segs)
        for i, s in enumerate(segs):
            a.seg[i] = s
        a.a0 = ptr(dz)
        if dz_bn is not None:
            if dz1 is None:
                raise L.TrunetHipError("weight gradient behind a BatchNorm needs the raw conv output next to dy")
            a.a_mode = PRO_BNBWD
            a.a1, a.ac0, a.ac1, a.ac2 = ptr(dz1), ptr(dz_bn.ca), ptr(dz_bn.cb), ptr(dz_bn.cc)
        else:
            a.a_mode = PRO_NONE
        a.w_numel = self._wg_total                      # image stride of the shared buffer
        a.w_partials = self._wg_slot(W)
        if bias is not None:
            a.b_partials = self._wg_slot(bias)
            a.b_stride, a.b_off = self._wg_total, b_off
        if PROFILE is not None:
            fl = 2.0 * N * M * sum(s.nchan * _seg_positions(s, 0, P) for s in segs)
            with _Timed("conv_wgrad_kernel<%s>" % ("true" if dz_bn is not None else "false"), fl,
                        "M%d K%s P%d" % (M, "+".join(str(s.nchan) for s in segs), P)):
                check(lib.trunet_conv_wgrad(a, L.stream()), "conv_wgrad")
        else:
            check(lib.trunet_conv_wgrad(a, L.stream()), "conv_wgrad")

    @staticmethod
    def _dz_seg(dy_, z_, bn_, C, Ln, **kw):
        if bn_ is None:
            return make_seg(dy_, C, Ln, mode=PRO_NONE, **kw)
        return make_seg(dy_, C, Ln, mode=PRO_BNBWD, src1=z_, c0=bn_.ca, c1=bn_.cb, c2=bn_.cc, **kw)

    def _pw_bwd(self, w, *, N, NP, P, M, dz, dz1, dz_bn, W, bias, segs, outs, grads, fused=True, a_m_off=0, w_m_off=0,
                b_off=0, must_fuse=False):
        """Backward of a Conv1d(k=1)+BatchNorm layer: weight/bias gradient and, per source segment, the data gradient with
        its ReLU mask / skip accumulation / BatchNorm-backward statistics.
        outs[i] = dict(out=tensor, src=Act or None (mask, + statistics when src.bn), accum=bool).
        One fused launch (trunet_pw_bwd) when the kernel supports the shape, else (TRUNET_ENOTSUP: e.g. tensors beyond
        its 32-bit row offsets, thin layers) trunet_conv_wgrad + one trunet_conv_gemm per source."""
        lib = L.lib()
        K = sum(s.nchan for s in segs)
        if fused:
            a = PwBwdArgs()
            aw = a.w
            aw.NP, aw.N, aw.P, aw.p_begin = NP, N, P, 0
            # a_m_off / w_m_off / b_off: a ROW BLOCK of a wider layer (the GRU input projection, 2 x 192 rows, as blocks of
            # 128 + 64): first dz row, first weight row inside W, first bias row
            aw.M, aw.a_L, aw.a_pos_off, aw.a_m_off = M, P, 0, a_m_off
            aw.ldw_m, aw.ldw_c, aw.w_m_off = K, 1, w_m_off
            aw.nseg = len(segs)
            aw.a0, aw.a1 = ptr(dz), ptr(dz1)
            aw.a_mode = PRO_BNBWD
            aw.ac0, aw.ac1, aw.ac2 = ptr(dz_bn.ca), ptr(dz_bn.cb), ptr(dz_bn.cc)
            aw.w_numel = self._wg_total                     # image stride of the shared buffer
            aw.w_partials, aw.b_partials = self._wg_slot(W), self._wg_slot(bias)
            aw.b_stride, aw.b_off = self._wg_total, b_off
            a.W = ptr(W.data)
            nparts = lib.trunet_pw_bwd_nparts()
            stat_parts = []
            for i, (sg, o) in enumerate(zip(segs, outs)):
                aw.seg[i] = sg
                d = a.dg[i]
                d.out = ptr(o["out"])
                fl = DG_STORE
                src = o.get("src")
                if src is not None:
                    fl |= DG_MASK
                    d.zmask = ptr(src.t)
                    if src.bn is not None:
                        fl |= DG_STATS
                        part = w.flat("pwb_partials%d" % i, nparts * sg.nchan * 2, zero=True)
                        d.e2, d.partials = ptr(src.bn.mean), ptr(part)
                        stat_parts.append((src.bn, "pwb_partials%d" % i))
                        if w.take_clean("pwb_partials%d" % i):
                            fl |= DG_PREZERO
                if o.get("accum"):
                    fl |= DG_ACCUM
                d.flags = fl
            if PROFILE is not None:
                fl_ = 4.0 * N * M * sum(s.nchan * _seg_positions(s, 0, P) for s in segs)
                # the instance trunet_pw_bwd launches, as rocprofv3 prints it: <AK, SEC, KSPLIT> (KSPLIT: two source row tiles)
                ksplit = K == 64 and os.environ.get("TRUNET_PWB_KSPLIT", "1") != "0"
                name = "pw_bwd_kernel<%d, %s, %s, %s>" % (16 if M <= 32 else (32 if M <= 64 else 64), "true" if K == 192 else "false",
                                                          "true" if ksplit else "false",
                                                          "true" if lib.trunet_gemm_x3_enable(-1) & L.X3_BWD else "false")
                # algorithmic bytes: dy and z_y once, every source row once, every gradient row written once (read as well
                # where it accumulates), fp32, valid frames only
                by_ = 4.0 * N * (2 * M * P + sum((3 if o.get("accum") else 2) * s.nchan * _seg_positions(s, 0, P)
                                                 for s, o in zip(segs, outs)))
                with _Timed(name, fl_, "M%d K%s P%d" % (M, "+".join(str(s.nchan) for s in segs), P), nbytes=by_):
                    rc = lib.trunet_pw_bwd(a, L.stream())
            else:
                rc = lib.trunet_pw_bwd(a, L.stream())
            if rc == 0:
                for bn, pname in stat_parts:
                    self._bn_bwd(w, bn, nparts, grads, part_name=pname)
                return
            for _, pname in stat_parts:          # nothing was launched: the statistics buffers are as clean as before
                w.pending.discard(pname)
            if rc != L.TRUNET_ENOTSUP or must_fuse:
                check(rc, "pw_bwd")
        assert not (a_m_off or w_m_off or b_off), "row blocks exist on the fused kernel only"
        # ---- separate launches
        self._wgrad(w, N=N, NP=NP, P=P, M=M, dz=dz, dz1=dz1, dz_L=P, dz_bn=dz_bn, W=W, ldw_m=K, ldw_c=1, segs=segs,
                    grads=grads, bias=bias)
        for sg, o in zip(segs, outs):
            src = o.get("src")
            p0, p1 = max(0, -sg.pos_off), min(P, sg.L - sg.pos_off)
            kw = {}
            if src is not None:
                if src.bn is not None:
                    kw = dict(zmask=src.t, e0=src.bn.scale, e1=src.bn.shift, e2=src.bn.mean, stats=sg.nchan)
                else:       # ReLU-only source (enc0): mask = z > 0
                    one = w.get("ones%d" % sg.nchan, (sg.nchan,))
                    zero = w.get("zeros%d" % sg.nchan, (sg.nchan,))
                    one.fill_(1.0)
                    zero.zero_()
                    kw = dict(zmask=src.t, e0=one, e1=zero, e2=zero)
            nparts = self._gemm(w, N=N, NP=NP, P=p1 - p0, p_begin=p0, M=sg.nchan, out=o["out"], out_L=sg.L,
                                out_pos_off=sg.pos_off, W=W.data, ldw_m=1, ldw_c=K, w_m_off=sg.woff,
                                segs=[self._dz_seg(dz, dz1, dz_bn, M, P)], epi=(EPI_ACCUM if o.get("accum") else 0), **kw)
            if src is not None and src.bn is not None:
                self._bn_bwd(w, src.bn, nparts, grads)

    # ---- block-level pieces of the backward schedule.  ``up`` = (dy, z, bn): the gradient at a layer's BatchNorm
    # output with the ReLU mask applied, that layer's raw conv output and its BatchNorm state (bn None: dy is the
    # gradient of the raw output itself).  Sources are Acts: with a BatchNorm state the data gradient is masked and its
    # BatchNorm-backward sums are reduced ("bn"); ``*_relu`` marks a ReLU-only source (enc0); a raw source (block
    # inputs, skip tensors) gets the plain gradient.
    def _bwd_tr(self, w, N, NP, ct, pw, a_pw, Lo, up, x1, x1_mask, skip, left, dy_x1, g_skip, dy_pw_name, grads):
        """FirstTrCNN / TrCNN / LastTrCNN (network.py:60-120): transposed conv, then the pointwise conv over
        [x1 (padded / cropped by ``left``) | skip].  x1_mask: x1 itself when its BatchNorm+ReLU sits between x1 and this
        block (full network), None for a raw block input."""
        dy, z, bn = up
        k, s_, pad = ct.kernel_size[0], ct.stride[0], ct.padding[0]
        Ci, Co = ct.in_channels, ct.out_channels
        dy_pw = w.get(dy_pw_name, (Ci, a_pw.L, NP))
        if not (FUSED_CONVT and bn is not None and self._convt_bwd(w, N, NP, ct, a_pw, Lo, dy, z, bn, dy_pw, grads)):
            # transposed conv: weight/bias gradient
            self._wgrad(w, N=N, NP=NP, P=Lo, M=Co, dz=dy, dz1=z, dz_L=Lo, dz_bn=bn, W=ct.weight, ldw_m=k,
                        ldw_c=Co * k, segs=[a_pw.seg(pos_off=pad - kk, woff=kk, pos_div=s_) for kk in range(k)],
                        grads=grads, bias=ct.bias)
            # transposed conv: data gradient -> dy of the pw BN (+ stats)
            segs = [self._dz_seg(dy, z, bn, Co, Lo, pos_mul=s_, pos_off=kk - pad, woff=kk) for kk in range(k)]
            nparts = self._gemm(w, N=N, NP=NP, P=a_pw.L, M=Ci, out=dy_pw, out_L=a_pw.L, W=ct.weight.data,
                                ldw_m=Co * k, ldw_c=k, segs=segs, zmask=a_pw.t, e0=a_pw.bn.scale, e1=a_pw.bn.shift,
                                e2=a_pw.bn.mean, stats=Ci)
            self._bn_bwd(w, a_pw.bn, nparts, grads)
        # pointwise conv over [x1 | skip]
        Lp = a_pw.L
        srcs = [x1.seg(pos_off=-left, woff=0)] + ([skip.seg(woff=x1.C)] if skip is not None else [])
        p0, p1 = max(0, left), min(Lp, x1.L + left)
        if p1 - p0 < x1.L:          # cropped positions of x1 (network.py:96-97 with a negative pad) get no gradient
            w.zero_crop(dy_x1, p0 - left, p1 - left)
        outs = [dict(out=dy_x1, src=x1_mask)] + ([dict(out=g_skip)] if skip is not None else [])
        # decoder.5's 8-row layer: trunet_pw_bwd pads its dz block to one 32-row MFMA tile (round 3; TRUNET_FUSED_THIN=0
        # keeps the three separate launches of rounds 1-2: conv_wgrad + two conv_gemm, 1.36 ms)
        fused = FUSED_PWBWD and (pw.out_channels % 32 == 0 or (pw.out_channels <= 32 and FUSED_THIN))
        self._pw_bwd(w, N=N, NP=NP, P=Lp, M=pw.out_channels, dz=dy_pw, dz1=a_pw.t, dz_bn=a_pw.bn, W=pw.weight,
                     bias=pw.bias, segs=srcs, outs=outs, grads=grads, fused=fused)

    def _convt_bwd(self, w, N, NP, ct, a_pw, Lo, dy, z, bn, dy_pw, grads):
        """Fused backward of ConvTranspose1d(64 -> 64) + BatchNorm (trunet_convt_bwd): weight / bias gradient, the data
        gradient at the pointwise BatchNorm's output (masked) and its BatchNorm-backward sums in one pass over (dy, z,
        source).  False when the kernel does not support the layer (TRUNET_ENOTSUP): the caller takes the separate launches."""
        lib = L.lib()
        k, s_, pad = ct.kernel_size[0], ct.stride[0], ct.padding[0]
        a = ConvtBwdArgs()
        a.NP, a.N, a.Lin, a.Lout, a.K, a.S, a.pad = NP, N, a_pw.L, Lo, k, s_, pad
        a.Ci, a.Co = ct.in_channels, ct.out_channels
        if a.Ci != 64 or a.Co != 64:
            return False
        nparts = lib.trunet_convt_bwd_nparts()
        part = w.flat("ct_partials", nparts * a.Ci * 2)
        a.dy, a.z = ptr(dy), ptr(z)
        a.ca, a.cb, a.cc = ptr(bn.ca), ptr(bn.cb), ptr(bn.cc)
        a.src, a.s_scale, a.s_shift, a.s_mean = ptr(a_pw.t), ptr(a_pw.bn.scale), ptr(a_pw.bn.shift), ptr(a_pw.bn.mean)
        a.W, a.dsrc, a.partials = ptr(ct.weight.data), ptr(dy_pw), ptr(part)
        a.w_numel = self._wg_total
        a.w_partials, a.b_partials = self._wg_slot(ct.weight), self._wg_slot(ct.bias)
        a.b_stride, a.b_off = self._wg_total, 0
        if PROFILE is not None:
            fl = 4.0 * N * a.Ci * a.Co * sum(1 for q in range(a_pw.L) for kk in range(k) if 0 <= q * s_ - pad + kk < Lo)
            x3 = lib.trunet_gemm_x3_enable(-1) & L.X3_BWD
            with _Timed("convt_bwd%s_kernel<%d, %d>" % ("_x3" if x3 else "", k, s_), fl, "L%d" % a_pw.L):
                rc = lib.trunet_convt_bwd(a, L.stream())
        else:
            rc = lib.trunet_convt_bwd(a, L.stream())
        if rc == L.TRUNET_ENOTSUP:
            return False
        check(rc, "convt_bwd")
        self._bn_bwd(w, a_pw.bn, nparts, grads, part_name="ct_partials")
        return True

    def _ident_bnbwd(self, w, C):
        """(ca, cb, cc) = (1, 0, 0): a dz operand that is already the gradient of the raw output, for kernels whose dz
        prologue is the BatchNorm backward ca dy + cb z + cc"""
        t = w.t.get("ident_bnbwd%d" % C)
        if t is None:
            import types
            ca = torch.ones(C, device=self._wg_base.device, dtype=torch.float32)
            zz = torch.zeros(C, device=self._wg_base.device, dtype=torch.float32)
            t = w.t["ident_bnbwd%d" % C] = types.SimpleNamespace(ca=ca, cb=zz, cc=zz.clone())
        return t

    def _bwd_fgru(self, w, N, NP, blk, up, hout, src, src_mask, dy_src, grads):
        """GRUBlock(128, 64, 64, True) (network.py:45-58): pointwise conv over hout, the recurrence, the input
        projection; dy_src receives the gradient of the block input ``src`` (masked + statistics when src_mask)."""
        lib, st = L.lib(), L.stream()
        dy, z, bn = up
        conv, gru = blk.conv[0], blk.GRU
        Hh, Lg = gru.hidden_size, hout.L
        dhout = w.get("dhout", (2 * Hh, Lg, NP))
        self._pw_bwd(w, N=N, NP=NP, P=Lg, M=conv.out_channels, dz=dy, dz1=z, dz_bn=bn, W=conv.weight,
                     bias=conv.bias, segs=[hout.seg()], outs=[dict(out=dhout)], grads=grads, fused=FUSED_PWBWD)
        # -------- GRU recurrence backward
        dgi = w.get("dgi", (6 * Hh, Lg, NP))
        dghn = w.get("dghn", (2 * Hh, Lg, NP))
        gates = w.t["gates"]
        check(lib.trunet_gru_bwd(ptr(dhout), ptr(hout.t), ptr(gates), ptr(gru.weight_hh_l0.data),
                                 ptr(gru.weight_hh_l0_reverse.data), ptr(dgi), ptr(dghn), Hh, Lg, NP, N, st), "gru_bwd")
        # (TRUNET_FUSED_GRU_PROJ=0: conv_wgrad + conv_gemm for the projection, rounds 1-2)
        fuse_proj = (FUSED_PWBWD and FUSED_GRU_PROJ and src_mask is not None and src_mask.bn is not None and
                     src_mask.t is src.t and 3 * Hh > 128 and (3 * Hh - 128) % 32 == 0 and gru.input_size % 32 == 0)
        for d, sfx in enumerate(("", "_reverse")):
            whh = getattr(gru, "weight_hh_l0" + sfx)
            bhh = getattr(gru, "bias_hh_l0" + sfx)
            wih_p = getattr(gru, "weight_ih_l0" + sfx)
            bih_p = getattr(gru, "bias_ih_l0" + sfx)
            hseg = make_seg(hout.t[d * Hh:(d + 1) * Hh], Hh, Lg, pos_off=(1 if d else -1))
            # rows 0..2H-1 (r, z) come from dgi, rows 2H..3H-1 (n) from dghn
            self._wgrad(w, N=N, NP=NP, P=Lg, M=2 * Hh, dz=dgi, dz_L=Lg, dz_bn=None, a_m_off=d * 3 * Hh, W=whh,
                        ldw_m=Hh, ldw_c=1, segs=[hseg], grads=grads, bias=bhh, b_off=0)
            self._wgrad(w, N=N, NP=NP, P=Lg, M=Hh, dz=dghn, dz_L=Lg, dz_bn=None, a_m_off=d * Hh, w_m_off=2 * Hh,
                        W=whh, ldw_m=Hh, ldw_c=1, segs=[hseg], grads=grads, bias=bhh, b_off=2 * Hh)
            if fuse_proj:
                continue
            # input projection weights: all 3H = 192 rows of dgi for this direction in one launch
            self._wgrad(w, N=N, NP=NP, P=Lg, M=3 * Hh, dz=dgi, dz_L=Lg, dz_bn=None, a_m_off=d * 3 * Hh, W=wih_p,
                        ldw_m=gru.input_size, ldw_c=1, segs=[src.seg()], grads=grads, bias=bih_p, b_off=0)
        if fuse_proj:
            # The input projection IS a pointwise layer (gi = W_ih a(src) + b_ih, 2 x 192 rows): its backward on the fused
            # kernel, one launch per row block (128 + 64 rows per direction; dz = dgi through the identity "BatchNorm
            # backward" ca = 1, cb = cc = 0).  Each launch writes its block of dW_ih / db_ih and ADDS its share of
            # W_ih^T dgi into dy_src under the source's ReLU mask (idempotent, so applying it at every launch is exact);
            # the BatchNorm-backward sums are taken by the last launch, from the finished gradient.  Replaces two
            # conv_wgrad launches and a conv_gemm that read the 384 dgi rows twice (K = 384 does not fit its LDS weight
            # block next to 128 output rows): 1.48 -> 1.25 ms.
            ident = self._ident_bnbwd(w, 6 * Hh)
            blocks = [(d, r0, m) for d in (0, 1) for (r0, m) in ((0, 128), (128, 3 * Hh - 128))]
            for i, (d, r0, m) in enumerate(blocks):
                sfx = "_reverse" if d else ""
                last = i == len(blocks) - 1
                o = dict(out=dy_src, src=(src_mask if last else Act(src_mask.t, src_mask.C, src_mask.L)), accum=i > 0)
                self._pw_bwd(w, N=N, NP=NP, P=Lg, M=m, dz=dgi, dz1=dgi, dz_bn=ident, W=getattr(gru, "weight_ih_l0" + sfx),
                             bias=getattr(gru, "bias_ih_l0" + sfx), segs=[src.seg()], outs=[o], grads=grads, fused=True,
                             a_m_off=d * 3 * Hh + r0, w_m_off=r0, b_off=r0, must_fuse=True)
            return
        # data gradient of the projection -> gradient of the block input (dy of enc5's BN in the full network)
        wih = w.t["wih"]
        kw = {}
        if src_mask is not None:
            kw = dict(zmask=src.t, e0=src.bn.scale, e1=src.bn.shift, e2=src.bn.mean, stats=src.C)
        nparts = self._gemm(w, N=N, NP=NP, P=Lg, M=src.C, out=dy_src, out_L=src.L, W=wih, ldw_m=1,
                            ldw_c=gru.input_size, segs=[make_seg(dgi, 6 * Hh, Lg)], **kw)
        if src_mask is not None:
            self._bn_bwd(w, src.bn, nparts, grads)

    def _bwd_dsc(self, w, N, NP, seq, a_pw, a_dw, up, prev, prev_mask, accum, dy_prev, dy_pw_name, grads):
        """DepthwiseSeparableConv1d (network.py:24-43): depthwise conv (fused dgrad + wgrad + BN-backward sums of the
        pointwise BatchNorm), then the pointwise conv; dy_prev receives the gradient of the block input ``prev``
        (added to the skip gradient already stored there when ``accum``)."""
        lib, st = L.lib(), L.stream()
        dy, z, bn = up
        pw, dwc = seq[0], seq[3]
        k, s_ = dwc.kernel_size[0], dwc.stride[0]
        C = dwc.out_channels
        dy_pw = w.get(dy_pw_name, (C, a_pw.L, NP))
        nparts = lib.trunet_dwconv_bwd_nparts(a_pw.L)
        part = w.flat("partials_dw", nparts * C * 2)
        wpart = w.flat("dw_w_partials", nparts * C * k)
        bpart = w.flat("dw_b_partials", nparts * C)
        tail = (ptr(bn.ca), ptr(bn.cb), ptr(bn.cc), ptr(a_pw.t), ptr(a_pw.bn.scale), ptr(a_pw.bn.shift), ptr(a_pw.bn.mean),
                ptr(dwc.weight.data), ptr(dy_pw), ptr(part), ptr(wpart), ptr(bpart), C, k, s_, a_pw.L, a_dw.L, NP, N, st)
        # z (the depthwise conv's own output) is recomputed from its input rows instead of read: one row pass less
        rc = lib.trunet_dwconv_bwd_rz(ptr(dy), ptr(dwc.bias.data), *tail) if DW_RZ else L.TRUNET_ENOTSUP
        if rc == L.TRUNET_ENOTSUP:
            rc = lib.trunet_dwconv_bwd(ptr(dy), ptr(z), *tail)
        check(rc, "dwconv_bwd")
        check(lib.trunet_reduce_partials(self._wg_slot(dwc.weight), ptr(wpart), nparts, C * k, 0, st), "reduce")
        check(lib.trunet_reduce_partials(self._wg_slot(dwc.bias), ptr(bpart), nparts, C, 0, st), "reduce")
        self._bn_bwd(w, a_pw.bn, nparts, grads, part_name="partials_dw")
        self._pw_bwd(w, N=N, NP=NP, P=a_pw.L, M=pw.out_channels, dz=dy_pw, dz1=a_pw.t, dz_bn=a_pw.bn, W=pw.weight,
                     bias=pw.bias, segs=[prev.seg()], outs=[dict(out=dy_prev, src=prev_mask, accum=accum)],
                     grads=grads, fused=FUSED_PWBWD)

    def _bwd_first(self, w, N, NP, c0, xa, dy, Lo, grads):
        """StandardConv1d (network.py:9-21): weight/bias gradient, one segment per tap of the input"""
        k, s_, pad = c0.kernel_size[0], c0.stride[0], c0.padding[0]
        segs = [make_seg(xa.t, xa.C, xa.L, pos_mul=s_, pos_off=kk - pad, woff=kk) for kk in range(k)]
        self._wgrad(w, N=N, NP=NP, P=Lo, M=c0.out_channels, dz=dy, dz_L=Lo, dz_bn=None, W=c0.weight,
                    ldw_m=xa.C * k, ldw_c=k, segs=segs, grads=grads, bias=c0.bias)

    def backward(self, ctx, gout):
        """gout: (N, 8, 257) cotangent.  Returns {parameter tensor: gradient}."""
        acts, N, NP, w, gen = ctx
        self._check_gen(w, gen)
        net = self.net
        lib = L.lib()
        st = L.stream()
        grads = {}
        self._wg_begin(w)
        gout = gout.contiguous()
        last = acts["dec5"]
        dyt = w.get("dy:dec5", (last.C, last.L, NP))
        check(lib.trunet_to_frames_last(ptr(gout), ptr(dyt), N, last.C, last.L, NP, st), "to_frames_last")

        # current upstream gradient: (dy tensor, z tensor, BN state of that z or None)
        up = (dyt, last.t, None)
        # -------- decoder, last to first
        for i in range(5, -1, -1):
            seq = (net.decoder[i].LastTrCNN if i == 5 else net.decoder[i].TrCNN) if i > 0 else net.decoder[0].FirstTrCNN
            if i > 0:
                x1, x1n = acts["dec%d" % (i - 1)], "dec%d" % (i - 1)
                skip = acts["enc%d" % (5 - i)]
                left = (skip.L - x1.L) // 2
                g_skip = w.get("dy:enc%d" % (5 - i), (skip.C, skip.L, NP))
            else:
                x1n = "tgru" if "tgru" in acts else "fgru"
                x1, skip, left, g_skip = acts[x1n], None, 0, None
            dy_x1 = w.get("dy:" + x1n, (x1.C, x1.L, NP))
            self._bwd_tr(w, N, NP, seq[3], seq[0], acts["dec%d.pw" % i], acts["dec%d" % i].L, up, x1, x1, skip, left,
                         dy_x1, g_skip, "dy:dec%d.pw" % i, grads)
            up = (dy_x1, x1.t, x1.bn)

        if "tgru.ctx" in acts:            # time-recurrent block between FGRU and the decoder
            up = self._tgru_seq_bwd(w, acts, up[0], up[2], N, NP, grads)

        # -------- FGRU
        enc5 = acts["enc5"]
        dy5 = w.get("dy:enc5", (enc5.C, enc5.L, NP))
        self._bwd_fgru(w, N, NP, net.FGRU, up, acts["hout"], enc5, enc5, dy5, grads)
        up = (dy5, enc5.t, enc5.bn)

        # -------- encoder 5..1: the data gradient accumulates onto the decoder's skip gradient stored in dy:enc{i-1}
        for i in range(5, 0, -1):
            prev = acts["enc%d" % (i - 1)]
            dy_prev = w.get("dy:enc%d" % (i - 1), (prev.C, prev.L, NP))
            self._bwd_dsc(w, N, NP, net.encoder[i].DepthwiseSeparableConv1d, acts["enc%d.pw" % i], acts["enc%d" % i], up,
                          prev, prev, True, dy_prev, "dy:enc%d.pw" % i, grads)
            up = (dy_prev, prev.t, prev.bn)

        # -------- first conv (its output is ReLU-only: up[0] is already masked)
        self._bwd_first(w, N, NP, net.encoder[0].StandardConv1d[0], acts["x"], up[0], acts["enc0"].L, grads)
        self._wg_finish(grads)
        return grads
