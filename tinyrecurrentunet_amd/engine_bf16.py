"""bf16 schedule of the TRU-Net body (BASELINE.json configs[2]; a build extension -- the reference has no reduced
precision path).

Same layer schedule as ``engine.TRUNetEngine`` (``/root/reference/network.py:122-171`` with repairs R1-R4), on the
``trunet_bf16_*`` kernels: every activation between layers and every activation gradient is stored as bf16 in the
octet layout ``[C/8][L][NP][8]`` and multiplied on ``v_mfma_f32_32x32x16_bf16``; accumulators, BatchNorm statistics and
coefficients, weight gradients (the flat fp32 gradient tensor of the fp32 engine, so the all-reduce and FusedAdamW are
unchanged) and the master weights stay fp32.  Of the frequency-recurrent bottleneck (FGRU) only the recurrence itself runs
on the fp32 kernels (fp32 gi / gates / hidden states); its input projection, pointwise conv and all their gradients are
bf16 GEMMs like the rest of the body.

What is not offered in bf16: the TGRU block and the stand-alone block classes (fp32 only).
"""
import ctypes as C

import torch

from . import _lib as L
from . import engine as E
import os

from ._lib import (DG_ACCUM, DG_MASK, DG_PREZERO, DG_STATS, DG_STORE, EPI_ACCUM, EPI_BIAS, EPI_F32OUT, EPI_MASK, EPI_PREZERO, EPI_RELU, EPI_STATS, PRO_BNBWD,
                   PRO_BNRELU, PRO_NONE, BConvtArgs, BGemmArgs, BPackDesc, BPwBwdArgs, BSeg, BWgradArgs, check, ptr, ptr16)
from .engine import BN_MOM, F_BINS, FRAME_PAD, Act, TRUNetEngine, _Timed, _seg_positions, ceil_to

BF16 = torch.bfloat16

# Backward of the pointwise convs: one fused launch (trunet_bf16_pw_bwd: weight gradient + every source's data gradient in one
# pass over dy, z, sources) instead of trunet_bf16_wgrad + one trunet_bf16_gemm per source; TRUNET_BF16_FUSED_PWBWD=0 keeps the
# separate launches (A/B measurements, and the reference the fused kernel is tested against bit for bit).
FUSED_PWBWD16 = os.environ.get("TRUNET_BF16_FUSED_PWBWD", "1") != "0"
# Backward of the 64 -> 64 transposed convs (decoder.0-4): one fused launch (trunet_bf16_convt_bwd) instead of trunet_bf16_wgrad
# + trunet_bf16_gemm over the tap segments; TRUNET_BF16_FUSED_CONVT=0 keeps the separate launches.
FUSED_CONVT16 = os.environ.get("TRUNET_BF16_FUSED_CONVT", "1") != "0"
# The FGRU input projection (M = 384), its data gradient and its weight gradients on the bf16 kernels (the recurrence, its
# W_hh gradients and the block's pointwise conv stay fp32); TRUNET_BF16_GRU_PROJ=0 keeps the whole block on the fp32 kernels.
GRU_PROJ16 = os.environ.get("TRUNET_BF16_GRU_PROJ", "1") != "0"
# the recurrence kernels read and write bf16 octets (trunet_bf16_gru_fwd / _bwd); "0": the fp32 kernels between conversions
GRU_IO16 = GRU_PROJ16 and os.environ.get("TRUNET_BF16_GRU_IO", "1") != "0"
# decoder.5's ConvTranspose1d(8 -> 8) on the fp32 thin-layer kernels (conv_smallm / wgrad_last): one octet per frame keeps the
# bf16 MFMA kernels at 0.4-1 TB/s there (forward 0.24 ms, weight gradient 0.46 ms, data gradient 0.22 ms against 0.14 / 0.16 /
# 0.14 ms in fp32 for twice the bytes), and the layer's output IS the fp32 module output.  "0": the bf16 kernels
LAST_CT32 = os.environ.get("TRUNET_BF16_LAST_CT32", "1") != "0"
# Every packed weight image of a step in one launch at the start of the forward (the plan is learnt during the first
# step); TRUNET_BF16_BATCH_PACK=0 packs in front of each GEMM instead.
BATCH_PACK16 = os.environ.get("TRUNET_BF16_BATCH_PACK", "1") != "0"
PACK_POOL_ELEMS = 4 << 20          # bf16 elements of the persistent image pool (a step uses ~1.5 M)


def bseg(src0, nchan, Ln, pos_mul=1, pos_off=0, pos_div=1, woff=0, mode=PRO_NONE, src1=None, c0=None, c1=None, c2=None):
    s = BSeg()
    s.src0, s.src1 = ptr16(src0), ptr16(src1)
    s.c0, s.c1, s.c2 = ptr(c0), ptr(c1), ptr(c2)
    s.nchan, s.L = nchan, Ln
    s.pos_mul, s.pos_off, s.pos_div = pos_mul, pos_off, pos_div
    s.woff, s.mode, s.kstep0 = woff, mode, 0
    return s


class Act16:
    """A bf16 activation in the octet layout [ceil(C/8)][L][NP][8] and how a consumer must read it."""

    def __init__(self, t, Cn, Ln, bn=None):
        self.t, self.C, self.L, self.bn = t, Cn, Ln, bn

    def seg(self, pos_off=0, woff=0, pos_mul=1, pos_div=1):
        if self.bn is None:
            return bseg(self.t, self.C, self.L, pos_mul, pos_off, pos_div, woff, PRO_NONE)
        return bseg(self.t, self.C, self.L, pos_mul, pos_off, pos_div, woff, PRO_BNRELU, c0=self.bn.scale,
                    c1=self.bn.shift)


def _oct(Cn):
    return (Cn + 7) // 8


def _ksteps(nchan):
    return (_oct(nchan) + 1) // 2


def _bgemm_name(M, segs, epi):
    """The template instance trunet_bf16_gemm launches (mirror of its dispatch), as rocprofv3 prints it"""
    pro = PRO_NONE
    for s in segs:
        if s.mode == PRO_BNBWD:
            pro = PRO_BNBWD
        elif s.mode == PRO_BNRELU and pro != PRO_BNBWD:
            pro = PRO_BNRELU
    full = M % 32 == 0 and all(s.nchan % 16 == 0 for s in segs)
    B, S, A, K = EPI_BIAS, EPI_STATS, EPI_ACCUM, EPI_MASK
    hot = {(PRO_BNRELU, B | S), (PRO_NONE, B | S), (PRO_BNBWD, K | S), (PRO_BNBWD, K | S | A), (PRO_BNBWD, K | A),
           (PRO_BNBWD, 0), (PRO_NONE, K | S)}
    if full and M == 128 and (pro, epi) in {(PRO_BNRELU, B | S), (PRO_NONE, B | S), (PRO_BNRELU, B | EPI_F32OUT)}:
        return "bgemm_kernel<4, %d, %d, true>" % (pro, epi)
    if full and M > 32 and (pro, epi) in hot:
        return "bgemm_kernel<2, %d, %d, true>" % (pro, epi)
    R = EPI_RELU
    thin1 = {(PRO_BNRELU, B | S), (PRO_BNRELU, B), (PRO_NONE, K | S)}
    thin2 = {(PRO_NONE, B | R), (PRO_BNBWD, K | S), (PRO_BNBWD, 0)}
    if not full and (pro, epi) in (thin1 if M <= 32 else thin2):
        return "bgemm_kernel<%d, %d, %d, false>" % (1 if M <= 32 else 2, pro, epi)
    return "bgemm_kernel<%d, -1, -1, false>" % (1 if M <= 32 else 2)


class TRUNetEngineBF16(TRUNetEngine):
    """TRUNetEngine with bf16 activation storage (see the module docstring)."""

    # ------------------------------------------------------------------ packed weight images
    def _pack_group(self, specs):
        """Several images back to back in the pool (the W^T row tiles of a pointwise layer with more than 128 source
        channels: one image per source, read by the kernel as ONE image): specs = [(W, M, ldw_m, ldw_c, w_m_off, nchan,
        woff), ...]; returns the address of the first.  The members are ordinary plan entries (tagged, so they are never
        shared with a stand-alone image of the same weight), allocated consecutively the first time and refreshed by the
        batched pack launch afterwards -- until round 4 these were packed by their own launches in every step."""
        plan = self.__dict__.setdefault("_pk", {})
        keys = [(W.data_ptr(), M, ldw_m, ldw_c, w_m_off, tuple(nchan), tuple(woff), "grp%d/%d" % (i, len(specs)))
                for i, (W, M, ldw_m, ldw_c, w_m_off, nchan, woff) in enumerate(specs)]
        have = [k in plan for k in keys]
        if any(have) and not all(have):             # a partly aged-out group: re-allocate all of it
            for k in keys:
                plan.pop(k, None)
            self._pkdirty = True
        if not all(have) and self.__dict__.get("_pkbuf") is not None:
            need = sum(((M + 31) // 32) * sum(_ksteps(c) for c in nchan) * 64 * 8 for (_, M, _, _, _, nchan, _) in specs)
            if self._pkoff + need > PACK_POOL_ELEMS:
                plan.clear()
                self._pkoff = 0
        ptrs = [self._pack(*sp, tag=k[-1]) for sp, k in zip(specs, keys)]
        for (W, M, ldw_m, ldw_c, w_m_off, nchan, woff), p0, p1 in zip(specs, ptrs, ptrs[1:]):
            assert p1 - p0 == 2 * ((M + 31) // 32) * sum(_ksteps(c) for c in nchan) * 64 * 8, "group images are not contiguous"
        return ptrs[0]

    def _pack(self, W, M, ldw_m, ldw_c, w_m_off, nchan, woff, tag=None):
        """Device address of the MFMA A-fragment image of (W, addressing): taken from this step's batch when the plan
        knows it, else packed now (and added to the plan)."""
        lib, st = L.lib(), L.stream()
        nks = sum(_ksteps(c) for c in nchan)
        key = (W.data_ptr(), M, ldw_m, ldw_c, w_m_off, tuple(nchan), tuple(woff)) + ((tag,) if tag else ())
        plan = self.__dict__.setdefault("_pk", {})
        if self.__dict__.get("_pkbuf") is None or self._pkbuf.device != W.device:
            self._pkbuf = torch.empty(PACK_POOL_ELEMS, device=W.device, dtype=BF16)
            self._pkoff, self._pkepoch, self._pkdirty = 0, 0, True
            plan.clear()
        e = plan.get(key)
        if e is not None:
            e["used"] = self._pkepoch
            if e["epoch"] == self._pkepoch:
                return e["ptr"]
        else:
            n = ((M + 31) // 32) * nks * 64 * 8
            if self._pkoff + n > PACK_POOL_ELEMS:        # pool exhausted (many shapes): start over, everything repacks
                plan.clear()
                self._pkoff = 0
            # Wt keeps the source storage alive while the plan can still launch a batched read of its address (the
            # optimizer re-points p.data at its first step; the entry of the old storage ages out one epoch later)
            e = dict(ptr=self._pkbuf.data_ptr() + 2 * self._pkoff, W=W.data_ptr(), Wt=W, M=M, ldw_m=ldw_m, ldw_c=ldw_c,
                     w_m_off=w_m_off, nchan=list(nchan), woff=list(woff), nks=nks, epoch=-1, used=self._pkepoch, elems=n)
            self._pkoff += (n + 63) // 64 * 64
            plan[key] = e
            self._pkdirty = True
        ns = len(nchan)
        rc = lib.trunet_bf16_pack_weight(W.data_ptr(), e["ptr"], M, ldw_m, ldw_c, w_m_off, ns, (C.c_int32 * ns)(*nchan),
                                         (C.c_int32 * ns)(*woff), st)
        if rc != nks:
            raise L.TrunetHipError("trunet_bf16_pack_weight: code %d (expected %d k-steps)" % (rc, nks))
        e["epoch"] = self._pkepoch
        return e["ptr"]

    def _pack_all(self):
        """Start of a forward: a new epoch; pack every image the plan holds in one launch."""
        if self.__dict__.get("_pkbuf") is None:
            return
        self._pkepoch += 1
        plan = self._pk
        stale = [k for k, e in plan.items() if e["used"] < self._pkepoch - 1]      # not used by the previous forward
        for k in stale:
            del plan[k]
            self._pkdirty = True
        if not BATCH_PACK16 or not plan:
            return
        if self._pkdirty:
            arr = (BPackDesc * len(plan))()
            for d, e in zip(arr, plan.values()):
                d.W, d.out = e["W"], e["ptr"]
                d.M, d.ldw_m, d.ldw_c, d.w_m_off, d.nseg, d.nks_total = e["M"], e["ldw_m"], e["ldw_c"], e["w_m_off"], \
                    len(e["nchan"]), e["nks"]
                k0 = 0
                for i, (cn, wo) in enumerate(zip(e["nchan"], e["woff"])):
                    d.nchan[i], d.woff[i], d.ks0[i] = cn, wo, k0
                    k0 += _ksteps(cn)
            host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self._pkdesc = host.to(self._pkbuf.device)
            self._pkmax = max(e["elems"] // 8 for e in plan.values())
            self._pkdirty = False
        check(L.lib().trunet_bf16_pack_weights_batch(self._pkdesc.data_ptr(), len(plan), self._pkmax, L.stream()),
              "bf16_pack_weights_batch")
        for e in plan.values():
            e["epoch"] = self._pkepoch

    # ------------------------------------------------------------------ launches
    def _get16(self, w, name, Cn, Ln, NP):
        return w.get(name, (_oct(Cn), Ln, NP, 8), dtype=BF16)

    def _gemm16(self, w, *, N, NP, P, M, out, out_L, W, ldw_m, ldw_c, segs, p_begin=0, out_pos_off=0, w_m_off=0, epi=0,
                bias=None, zmask=None, e0=None, e1=None, e2=None, stats=None, m_out_off=0):
        """trunet_bf16_gemm: out[m][p + out_pos_off] = epi(sum_seg W_seg . pro(src_seg)); the fp32 weight is packed into
        MFMA A fragments first (trunet_bf16_pack_weight, a few microseconds)."""
        lib, st = L.lib(), L.stream()
        nseg = len(segs)
        nchan = (C.c_int32 * nseg)(*[s.nchan for s in segs])
        woff = (C.c_int32 * nseg)(*[s.woff for s in segs])
        nks = sum(_ksteps(s.nchan) for s in segs)
        wfrag_ptr = self._pack(W, M, ldw_m, ldw_c, w_m_off, [s.nchan for s in segs], [s.woff for s in segs])
        a = BGemmArgs()
        a.NP, a.N, a.P, a.p_begin = NP, N, P, p_begin
        a.M, a.out_L, a.out_pos_off, a.nseg, a.nks_total = M, out_L, out_pos_off, nseg, nks
        k0 = 0
        for i, s in enumerate(segs):
            s.kstep0 = k0
            k0 += _ksteps(s.nchan)
            a.seg[i] = s
        a.out, a.wfrag = (ptr(out) if epi & EPI_F32OUT else ptr16(out)), wfrag_ptr
        a.m_out_off = m_out_off
        if bias is not None:
            epi |= EPI_BIAS
            a.bias = ptr(bias)
        if zmask is not None:
            epi |= EPI_MASK
            a.zmask, a.e0, a.e1, a.e2 = ptr16(zmask), ptr(e0), ptr(e1), ptr(e2)
        nparts = 0
        if stats is not None:
            epi |= EPI_STATS
            nparts = lib.trunet_bf16_gemm_nparts()
            part = w.flat("partials", nparts * stats * 2, zero=True)
            a.partials, a.M_stat = ptr(part), stats
        a.epi = epi | (EPI_PREZERO if (stats is not None and w.take_clean("partials")) else 0)
        if E.PROFILE is not None:
            # algorithmic bytes: every valid source row read once (twice for a BatchNorm-backward pair), the output row
            # written once (+ read for accumulate / mask), 2 bytes per element over the N valid frames
            by = 0
            for s in segs:
                by += 2 * s.nchan * _seg_positions(s, p_begin, P) * (2 if s.mode == PRO_BNBWD else 1)
            by += 2 * M * P * (1 + (1 if epi & EPI_ACCUM else 0) + (1 if epi & EPI_MASK else 0))
            tag = "M%d K%s P%d" % (M, "+".join(str(s.nchan) for s in segs), P)
            with _Timed(_bgemm_name(M, segs, epi), float(by) * N, tag):
                check(lib.trunet_bf16_gemm(a, st), "bf16_gemm")
            return nparts
        check(lib.trunet_bf16_gemm(a, st), "bf16_gemm")
        return nparts

    def _wgrad16(self, w, *, N, NP, P, M, dz, dz_L, dz_bn, W, ldw_m, ldw_c, segs, grads, bias=None, a_pos_off=0,
                 w_m_off=0, b_off=0, dz1=None):
        lib = L.lib()
        a = BWgradArgs()
        a.NP, a.N, a.P, a.p_begin = NP, N, P, 0
        a.M, a.a_L, a.a_pos_off = M, dz_L, a_pos_off
        a.ldw_m, a.ldw_c, a.w_m_off = ldw_m, ldw_c, w_m_off
        a.nseg = len(segs)
        for i, s in enumerate(segs):
            a.seg[i] = s
        a.a0 = ptr16(dz)
        if dz_bn is not None:
            a.a_mode = PRO_BNBWD
            a.a1, a.ac0, a.ac1, a.ac2 = ptr16(dz1), ptr(dz_bn.ca), ptr(dz_bn.cb), ptr(dz_bn.cc)
        else:
            a.a_mode = PRO_NONE
        a.w_numel = self._wg_total
        a.w_partials = self._wg_slot(W)
        if bias is not None:
            a.b_partials = self._wg_slot(bias)
            a.b_stride, a.b_off = self._wg_total, b_off
        if E.PROFILE is not None:
            by = 2 * M * P * (2 if dz_bn is not None else 1) + sum(2 * s.nchan * _seg_positions(s, 0, P) for s in segs)
            with _Timed("bwgrad_kernel<%s, %d, false>" % ("true" if dz_bn is not None else "false",
                                                           1 if any(s.mode == PRO_BNRELU for s in segs) else 0), float(by) * N, "M%d K%s P%d" % (M, "+".join(str(s.nchan) for s in segs), P)):
                check(lib.trunet_bf16_wgrad(a, L.stream()), "bf16_wgrad")
        else:
            check(lib.trunet_bf16_wgrad(a, L.stream()), "bf16_wgrad")

    # ------------------------------------------------------------------ forward layers
    def _pw(self, w, name, srcs, conv, bn, N, NP, training, x1_left=0):
        Ln = srcs[-1].L
        Co, K = conv.out_channels, conv.in_channels
        out = self._get16(w, "z:" + name, Co, Ln, NP)
        segs, off = [], 0
        for i, s in enumerate(srcs):
            segs.append(s.seg(pos_off=(-x1_left if (i == 0 and len(srcs) == 2) else 0), woff=off))
            off += s.C
        assert off == K
        nparts = self._gemm16(w, N=N, NP=NP, P=Ln, M=Co, out=out, out_L=Ln, W=conv.weight.data, ldw_m=K, ldw_c=1,
                              segs=segs, bias=conv.bias.data, stats=(Co if (bn is not None and training) else None))
        st = self._bn_fwd(w, name, bn, Co, N * Ln, nparts, training) if bn is not None else None
        return Act16(out, Co, Ln, st)

    def _convT(self, w, name, src, conv, bn, N, NP, training):
        k, s, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
        Co = conv.out_channels
        Lo = (src.L - 1) * s - 2 * pad + k
        out = self._get16(w, "z:" + name, Co, Lo, NP)
        segs = [src.seg(pos_off=pad - kk, woff=kk, pos_div=s) for kk in range(k)]
        nparts = self._gemm16(w, N=N, NP=NP, P=Lo, M=Co, out=out, out_L=Lo, W=conv.weight.data, ldw_m=k, ldw_c=Co * k,
                              segs=segs, bias=conv.bias.data, stats=(Co if (bn is not None and training) else None))
        st = self._bn_fwd(w, name, bn, Co, N * Lo, nparts, training) if bn is not None else None
        return Act16(out, Co, Lo, st)

    def _dw(self, w, name, src, conv, bn, N, NP, training):
        k, s = conv.kernel_size[0], conv.stride[0]
        Cn = conv.out_channels
        Lo = (src.L + 2 * (k // 2) - k) // s + 1
        out = self._get16(w, "z:" + name, Cn, Lo, NP)
        lib = L.lib()
        nparts = lib.trunet_bf16_dw_nparts(NP, Lo)
        part = w.flat("partials_dw", nparts * Cn * 2)
        if E.PROFILE is not None:
            with _Timed("bdw_fwd_kernel<%d, %d>" % (k, s), 2.0 * Cn * (src.L + Lo) * N, "L%d" % Lo):
                check(lib.trunet_bf16_dwconv_fwd(ptr16(src.t), ptr(src.bn.scale), ptr(src.bn.shift), ptr(conv.weight.data),
                                                 ptr(conv.bias.data), ptr16(out), ptr(part), Cn, k, s, src.L, Lo, NP, N,
                                                 L.stream()), "bf16_dwconv_fwd")
        else:
            check(lib.trunet_bf16_dwconv_fwd(ptr16(src.t), ptr(src.bn.scale), ptr(src.bn.shift), ptr(conv.weight.data),
                                             ptr(conv.bias.data), ptr16(out), ptr(part), Cn, k, s, src.L, Lo, NP, N,
                                             L.stream()), "bf16_dwconv_fwd")
        st = w.bn(name, Cn)
        st.module, st.count = bn, float(N * Lo)
        if training:
            L.bump_mutation_epoch()     # running statistics are written through raw pointers
            rm = bn.running_mean if bn.track_running_stats else None
            rv = bn.running_var if bn.track_running_stats else None
            mom = BN_MOM if bn.momentum is None else bn.momentum
            nbt = bn.num_batches_tracked.data_ptr() if bn.track_running_stats else None
            check(lib.trunet_bn_finalize_fwd(ptr(part), nparts, Cn, float(N * Lo), ptr(bn.weight.data), ptr(bn.bias.data),
                                             bn.eps, mom, ptr(rm), ptr(rv), ptr(st.scale), ptr(st.shift), ptr(st.mean),
                                             ptr(st.rstd), nbt, L.stream()), "bn_finalize_fwd")
        else:
            check(lib.trunet_bn_eval_affine(Cn, ptr(bn.weight.data), ptr(bn.bias.data), ptr(bn.running_mean),
                                            ptr(bn.running_var), bn.eps, ptr(st.scale), ptr(st.shift), L.stream()),
                  "bn_eval_affine")
        return Act16(out, Cn, Lo, st)

    def _gru16(self, w, cur, gru, N, NP, training):
        """engine._gru with the input projection (both directions, 6H = 384 rows) on the bf16 GEMM: three launches of 128
        rows that write fp32 frames-last gi for the fp32 recurrence kernel"""
        lib = L.lib()
        Hh = gru.hidden_size
        wih, bih = w.t["wih"], w.t["bih"]       # concatenated by forward() BEFORE the batched weight pack read them
        Lg = cur.L
        if GRU_IO16:
            # gi, the recurrence output and the saved gates as octets: no fp32 tensor and no conversion launch on this path
            gi16 = self._get16(w, "gi16", 6 * Hh, Lg, NP)
            for m0 in range(0, 6 * Hh, 128):
                self._gemm16(w, N=N, NP=NP, P=Lg, M=min(128, 6 * Hh - m0), out=gi16[m0 // 8:], out_L=Lg, W=wih,
                             ldw_m=gru.input_size, ldw_c=1, segs=[cur.seg()], bias=bih[m0:], w_m_off=m0)
            hout16 = self._get16(w, "hout16", 2 * Hh, Lg, NP)
            gates16 = self._get16(w, "gates16", 8 * Hh, Lg, NP) if training else None      # [dir][r, z, n, gh][Hh]
            check(lib.trunet_bf16_gru_fwd(ptr16(gi16), ptr(gru.weight_hh_l0.data), ptr(gru.bias_hh_l0.data),
                                          ptr(gru.weight_hh_l0_reverse.data), ptr(gru.bias_hh_l0_reverse.data), ptr16(hout16),
                                          ptr16(gates16) if training else None, Hh, Lg, NP, L.stream()), "bf16_gru_fwd")
            return Act16(hout16, 2 * Hh, Lg)
        gi = w.get("gi", (6 * Hh, Lg, NP))
        for m0 in range(0, 6 * Hh, 128):
            self._gemm16(w, N=N, NP=NP, P=Lg, M=min(128, 6 * Hh - m0), out=gi, out_L=Lg, W=wih, ldw_m=gru.input_size, ldw_c=1,
                         segs=[cur.seg()], bias=bih, epi=EPI_F32OUT, w_m_off=m0, m_out_off=m0)
        hout = w.get("hout", (2 * Hh, Lg, NP))
        gates = w.get("gates", (2, 4, Hh, Lg, NP)) if training else None
        check(lib.trunet_gru_fwd(ptr(gi), ptr(gru.weight_hh_l0.data), ptr(gru.bias_hh_l0.data),
                                 ptr(gru.weight_hh_l0_reverse.data), ptr(gru.bias_hh_l0_reverse.data), ptr(hout),
                                 ptr(gates), Hh, Lg, NP, L.stream()), "gru_fwd")
        return Act(hout, 2 * Hh, Lg)

    def _bwd_fgru16(self, w, N, NP, blk, up, hout, hout16, src16, dy_src16, grads):
        """engine._bwd_fgru with the input projection's data gradient and weight gradients on the bf16 kernels: dgi is
        converted to octets once and feeds the W_ih gradients (trunet_bf16_wgrad, sources = enc5's activation) and
        dy(enc5) = W_ih^T dgi with ReLU mask and BatchNorm-backward sums (trunet_bf16_gemm, 24 k-steps)."""
        lib, st = L.lib(), L.stream()
        dy, z, bn = up
        conv, gru = blk.conv[0], blk.GRU
        Hh, Lg = gru.hidden_size, hout.L
        # the block's pointwise conv + BatchNorm (bf16, fused); its data gradient goes to the fp32 recurrence kernel as fp32
        dhout16 = self._get16(w, "dhout16", 2 * Hh, Lg, NP)
        self._pw_bwd16(w, N=N, NP=NP, P=Lg, M=conv.out_channels, dz=dy, dz1=z, dz_bn=bn, W=conv.weight, bias=conv.bias,
                       segs=[hout16.seg()], outs=[dict(out=dhout16)], grads=grads)
        if GRU_IO16:
            dgi16 = self._get16(w, "dgi16", 6 * Hh, Lg, NP)
            dghn16 = self._get16(w, "dghn16", 2 * Hh, Lg, NP)
            check(lib.trunet_bf16_gru_bwd(ptr16(dhout16), ptr16(hout16.t), ptr16(w.t["gates16"]), ptr(gru.weight_hh_l0.data),
                                          ptr(gru.weight_hh_l0_reverse.data), ptr16(dgi16), ptr16(dghn16), Hh, Lg, NP, st),
                  "bf16_gru_bwd")
        else:
            dhout = self._to32(w, "dhout", dhout16, 2 * Hh, Lg, NP)
            dgi = w.get("dgi", (6 * Hh, Lg, NP))
            dghn = w.get("dghn", (2 * Hh, Lg, NP))
            gates = w.t["gates"]
            check(lib.trunet_gru_bwd(ptr(dhout), ptr(hout.t), ptr(gates), ptr(gru.weight_hh_l0.data),
                                     ptr(gru.weight_hh_l0_reverse.data), ptr(dgi), ptr(dghn), Hh, Lg, NP, N, st), "gru_bwd")
            dgi16 = self._to16(w, "dgi16", dgi, 6 * Hh, Lg, NP)
            dghn16 = self._to16(w, "dghn16", dghn, 2 * Hh, Lg, NP)
        for d, sfx in enumerate(("", "_reverse")):
            whh = getattr(gru, "weight_hh_l0" + sfx)
            bhh = getattr(gru, "bias_hh_l0" + sfx)
            wih_p = getattr(gru, "weight_ih_l0" + sfx)
            bih_p = getattr(gru, "bias_ih_l0" + sfx)
            # recurrent weights: h_{t-1} of this direction (the neighbouring position) against the (r, z) rows of dgi and the
            # n rows of dghn
            hseg = bseg(hout16.t[d * Hh // 8:(d + 1) * Hh // 8], Hh, Lg, pos_off=(1 if d else -1))
            self._wgrad16(w, N=N, NP=NP, P=Lg, M=2 * Hh, dz=dgi16[d * 3 * Hh // 8:], dz_L=Lg, dz_bn=None, W=whh, ldw_m=Hh,
                          ldw_c=1, segs=[hseg], grads=grads, bias=bhh, b_off=0)
            hseg = bseg(hout16.t[d * Hh // 8:(d + 1) * Hh // 8], Hh, Lg, pos_off=(1 if d else -1))
            self._wgrad16(w, N=N, NP=NP, P=Lg, M=Hh, dz=dghn16[d * Hh // 8:], dz_L=Lg, dz_bn=None, W=whh, ldw_m=Hh,
                          ldw_c=1, segs=[hseg], grads=grads, bias=bhh, w_m_off=2 * Hh, b_off=2 * Hh)
            # input projection weights: rows [0, 128) and [128, 192) of this direction's 3H = 192 rows of dgi
            for r0, M in ((0, 128), (128, 3 * Hh - 128)):
                oct0 = (d * 3 * Hh + r0) // 8
                self._wgrad16(w, N=N, NP=NP, P=Lg, M=M, dz=dgi16[oct0:], dz_L=Lg, dz_bn=None, W=wih_p,
                              ldw_m=gru.input_size, ldw_c=1, segs=[src16.seg()], grads=grads, bias=bih_p, w_m_off=r0, b_off=r0)
        wih = w.t["wih"]
        nparts = self._gemm16(w, N=N, NP=NP, P=Lg, M=src16.C, out=dy_src16, out_L=src16.L, W=wih, ldw_m=1,
                              ldw_c=gru.input_size, segs=[bseg(dgi16, 6 * Hh, Lg)], zmask=src16.t, e0=src16.bn.scale,
                              e1=src16.bn.shift, e2=src16.bn.mean, stats=src16.C)
        self._bn_bwd(w, src16.bn, nparts, grads)

    def _to16(self, w, name, t32, Cn, Ln, NP):
        t16 = self._get16(w, name, Cn, Ln, NP)
        check(L.lib().trunet_bf16_from_frames_last(ptr(t32), ptr16(t16), Cn, Ln, NP, L.stream()), "bf16_from_frames_last")
        return t16

    def _to32(self, w, name, t16, Cn, Ln, NP):
        t32 = w.get(name, (Cn, Ln, NP))
        check(L.lib().trunet_bf16_to_frames_last(ptr16(t16), ptr(t32), Cn, Ln, NP, L.stream()), "bf16_to_frames_last")
        return t32

    # ------------------------------------------------------------------ stand-alone blocks: fp32 only
    def block_forward(self, *a, **k):
        raise L.TrunetHipError("the stand-alone block classes run in fp32 only")

    gru_block_forward = block_forward

    # ------------------------------------------------------------------ forward
    def forward(self, x, training, tgru_state=None, tgru_T=None, record=False):
        if tgru_state is not None or tgru_T is not None:
            raise L.TrunetHipError("the TGRU block is not part of the bf16 schedule (fp32 only)")
        net = self.net
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 3 and x.shape[2] == F_BINS
        x = x.contiguous()
        N, Cin = x.shape[0], x.shape[1]
        NP = ceil_to(N, FRAME_PAD)
        w = self.ws(NP, x.device, record)
        if record:
            w.gen += 1
        lib, st = L.lib(), L.stream()
        acts = {}
        if GRU_PROJ16:
            # both directions' W_ih / b_ih as one 384-row operand.  This workspace tensor is a SOURCE of the batched weight pack
            # below, so it must hold this step's weights before that launch reads it (until round 3 the concatenation sat in
            # _gru16, after the pack: from the second step on the projection ran on the weights of the step before)
            gru = net.FGRU.GRU
            Hh = gru.hidden_size
            torch.cat((gru.weight_ih_l0.data, gru.weight_ih_l0_reverse.data), 0, out=w.get("wih", (6 * Hh, gru.input_size)))
            torch.cat((gru.bias_ih_l0.data, gru.bias_ih_l0_reverse.data), 0, out=w.get("bih", (6 * Hh,)))
        self._pack_all()

        x16t = self._get16(w, "x16", Cin, F_BINS, NP)
        check(lib.trunet_bf16_from_ncl(ptr(x), ptr16(x16t), N, Cin, F_BINS, NP, st), "bf16_from_ncl")
        x16 = Act16(x16t, Cin, F_BINS)
        acts["x"] = x16
        # first conv (network.py:13): one segment per tap of the strided input, bias + ReLU in the epilogue
        c0 = net.encoder[0].StandardConv1d[0]
        k, s_, pad = c0.kernel_size[0], c0.stride[0], c0.padding[0]
        assert c0.in_channels == Cin
        L0 = (F_BINS + 2 * pad - k) // s_ + 1
        a0 = self._get16(w, "z:enc0", c0.out_channels, L0, NP)
        self._gemm16(w, N=N, NP=NP, P=L0, M=c0.out_channels, out=a0, out_L=L0, W=c0.weight.data, ldw_m=Cin * k, ldw_c=k,
                     segs=[x16.seg(pos_mul=s_, pos_off=kk - pad, woff=kk) for kk in range(k)], bias=c0.bias.data,
                     epi=EPI_RELU)
        cur = acts["enc0"] = Act16(a0, c0.out_channels, L0)
        for i in range(1, 6):
            seq = net.encoder[i].DepthwiseSeparableConv1d
            cur = acts["enc%d.pw" % i] = self._pw(w, "enc%d.pw" % i, [cur], seq[0], seq[1], N, NP, training)
            cur = acts["enc%d" % i] = self._dw(w, "enc%d" % i, cur, seq[3], seq[4], N, NP, training)

        # FGRU: the recurrence and the block's pointwise conv in fp32
        if GRU_PROJ16:
            hout = acts["hout"] = self._gru16(w, cur, net.FGRU.GRU, N, NP, training)
            if GRU_IO16:
                h16 = acts["hout16"] = hout
            else:   # the block's pointwise conv on the bf16 kernels as well: the recurrence output once more as octets
                h16 = acts["hout16"] = Act16(self._to16(w, "hout16", hout.t, hout.C, hout.L, NP), hout.C, hout.L)
            cur = acts["fgru"] = self._pw(w, "fgru", [h16], net.FGRU.conv[0], net.FGRU.conv[1], N, NP, training)
        else:
            enc5f = acts["enc5.f32"] = Act(self._to32(w, "z:enc5.f32", cur.t, cur.C, cur.L, NP), cur.C, cur.L, cur.bn)
            acts["hout"] = self._gru(w, enc5f, net.FGRU.GRU, N, NP, training)
            fg = acts["fgru.f32"] = TRUNetEngine._pw(self, w, "fgru", [acts["hout"]], net.FGRU.conv[0], net.FGRU.conv[1], N,
                                                     NP, training)
            cur = acts["fgru"] = Act16(self._to16(w, "z:fgru16", fg.t, fg.C, fg.L, NP), fg.C, fg.L, fg.bn)

        seq = net.decoder[0].FirstTrCNN
        cur = acts["dec0.pw"] = self._pw(w, "dec0.pw", [cur], seq[0], seq[1], N, NP, training)
        cur = acts["dec0"] = self._convT(w, "dec0", cur, seq[3], seq[4], N, NP, training)
        for i in range(1, 6):
            skip = acts["enc%d" % (5 - i)]
            seq = net.decoder[i].TrCNN if i < 5 else net.decoder[i].LastTrCNN
            left = (skip.L - cur.L) // 2
            cur = acts["dec%d.pw" % i] = self._pw(w, "dec%d.pw" % i, [cur, skip], seq[0], seq[1], N, NP, training,
                                                  x1_left=left)
            if i == 5 and LAST_CT32:
                # the stored (rounded) pointwise output as fp32, same BatchNorm state: the transposed conv and its output in fp32
                pw32 = acts["dec5.pw32"] = Act(self._to32(w, "z:dec5.pw32", cur.t, cur.C, cur.L, NP), cur.C, cur.L, cur.bn)
                cur = acts["dec5"] = TRUNetEngine._convT(self, w, "dec5.f32", pw32, seq[3], None, N, NP, training)
                break
            cur = acts["dec%d" % i] = self._convT(w, "dec%d" % i, cur, seq[3], seq[4] if i < 5 else None, N, NP, training)
        out = torch.empty((N, cur.C, cur.L), device=x.device, dtype=torch.float32)
        if isinstance(cur, Act16):
            check(lib.trunet_bf16_to_ncl(ptr16(cur.t), ptr(out), N, cur.C, cur.L, NP, st), "bf16_to_ncl")
        else:
            check(lib.trunet_from_frames_last(ptr(cur.t), ptr(out), N, cur.C, cur.L, NP, st), "from_frames_last")
        return out, (acts, N, NP, w, w.gen)

    # ------------------------------------------------------------------ backward pieces
    @staticmethod
    def _dz_seg16(dy_, z_, bn_, Cn, Ln, **kw):
        if bn_ is None:
            return bseg(dy_, Cn, Ln, mode=PRO_NONE, **kw)
        return bseg(dy_, Cn, Ln, mode=PRO_BNBWD, src1=z_, c0=bn_.ca, c1=bn_.cb, c2=bn_.cc, **kw)

    def _pw_bwd16(self, w, *, N, NP, P, M, dz, dz1, dz_bn, W, bias, segs, outs, grads):
        """Backward of a Conv1d(k=1)+BatchNorm layer: one weight-gradient launch, one data-gradient launch per source
        (ReLU mask / skip accumulation / BatchNorm-backward statistics in its epilogue)."""
        K = sum(s.nchan for s in segs)
        if FUSED_PWBWD16 and dz_bn is not None and self._pw_bwd16_fused(w, N, NP, P, M, dz, dz1, dz_bn, W, bias, segs, outs,
                                                                        grads, K):
            return
        self._wgrad16(w, N=N, NP=NP, P=P, M=M, dz=dz, dz1=dz1, dz_L=P, dz_bn=dz_bn, W=W, ldw_m=K, ldw_c=1, segs=segs,
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
            nparts = self._gemm16(w, N=N, NP=NP, P=p1 - p0, p_begin=p0, M=sg.nchan, out=o["out"], out_L=sg.L,
                                  out_pos_off=sg.pos_off, W=W.data, ldw_m=1, ldw_c=K, w_m_off=sg.woff,
                                  segs=[self._dz_seg16(dz, dz1, dz_bn, M, P)], epi=(EPI_ACCUM if o.get("accum") else 0),
                                  **kw)
            if src is not None and src.bn is not None:
                self._bn_bwd(w, src.bn, nparts, grads)

    def _pw_bwd16_fused(self, w, N, NP, P, M, dz, dz1, dz_bn, W, bias, segs, outs, grads, K):
        """trunet_bf16_pw_bwd; False when the kernel does not take the shape (thin layers: the caller runs the separate
        launches)"""
        lib, st = L.lib(), L.stream()
        if M % 16 or M > 128 or any(s.nchan % 32 or s.pos_mul != 1 or s.pos_div != 1 for s in segs):
            return False
        if sum(s.nchan for s in segs) > 192:
            return False
        nks = _ksteps(M)
        nrt_total = sum(s.nchan // 32 for s in segs)
        # W^T of all sources as ONE image: A(c, m) = W[m*K + c] for c over the concatenated source channels (the sources'
        # woff are consecutive: woff_s = channels before s), i.e. row tiles of the sources one after the other
        assert [s.woff for s in segs] == [sum(t.nchan for t in segs[:i]) for i in range(len(segs))]
        if K <= 128:
            wfragT_ptr = self._pack(W.data, K, 1, K, 0, [M], [0])
        else:       # more than 128 rows (the pack kernel's limit per image): one image per source, back to back in the pool
            wfragT_ptr = self._pack_group([(W.data, s.nchan, 1, K, s.woff, [M], [0]) for s in segs])
        a = BPwBwdArgs()
        aw = a.w
        aw.NP, aw.N, aw.P, aw.p_begin = NP, N, P, 0
        aw.M, aw.a_L, aw.a_pos_off, aw.a_mode = M, P, 0, PRO_BNBWD
        aw.ldw_m, aw.ldw_c, aw.w_m_off, aw.nseg = K, 1, 0, len(segs)
        aw.a0, aw.a1 = ptr16(dz), ptr16(dz1)
        aw.ac0, aw.ac1, aw.ac2 = ptr(dz_bn.ca), ptr(dz_bn.cb), ptr(dz_bn.cc)
        aw.w_numel = self._wg_total
        aw.w_partials, aw.b_partials = self._wg_slot(W), self._wg_slot(bias)
        aw.b_stride, aw.b_off = self._wg_total, 0
        a.dg.wfragT, a.dg.nrt_total = wfragT_ptr, nrt_total
        nparts = lib.trunet_bf16_pw_bwd_nparts()
        stat_parts = []
        for i, (sg, o) in enumerate(zip(segs, outs)):
            aw.seg[i] = sg
            a.dg.out[i] = ptr16(o["out"])
            fl = DG_STORE
            src = o.get("src")
            if src is not None:
                fl |= DG_MASK
                if src.bn is not None:
                    fl |= DG_STATS
                    part = w.flat("pwb16_partials%d" % i, nparts * sg.nchan * 2, zero=True)
                    a.dg.mean[i], a.dg.partials[i] = ptr(src.bn.mean), ptr(part)
                    stat_parts.append((src.bn, "pwb16_partials%d" % i))
                    if w.take_clean("pwb16_partials%d" % i):
                        fl |= DG_PREZERO
            if o.get("accum"):
                fl |= DG_ACCUM
            a.dg.flags[i] = fl
        if E.PROFILE is not None:
            by = 2 * M * P * 2
            for sg, o in zip(segs, outs):
                npos = _seg_positions(sg, 0, P)
                by += 2 * sg.nchan * npos * (2 + (1 if o.get("accum") else 0))       # source read, gradient written (+ read)
            with _Timed("bwgrad_kernel<true, %d, true>" % (1 if any(s.mode == PRO_BNRELU for s in segs) else 0),
                        float(by) * N, "M%d K%s P%d" % (M, "+".join(str(s.nchan) for s in segs), P)):
                rc = lib.trunet_bf16_pw_bwd(a, st)
        else:
            rc = lib.trunet_bf16_pw_bwd(a, st)
        if rc == L.TRUNET_ENOTSUP:
            for _, pname in stat_parts:          # nothing was launched: the statistics buffers are as clean as before
                w.pending.discard(pname)
            return False
        check(rc, "bf16_pw_bwd")
        for bn, pname in stat_parts:
            self._bn_bwd(w, bn, nparts, grads, part_name=pname)
        return True

    def _convt_bwd16(self, w, N, NP, ct, a_pw, Lo, dy, z, bn, dy_pw, grads):
        """trunet_bf16_convt_bwd: weight / bias gradient, the masked data gradient at the pointwise BatchNorm's output and its
        BatchNorm-backward sums in one pass over (dy, z, source).  False when the kernel does not take the layer."""
        lib, st = L.lib(), L.stream()
        k, s_, pad = ct.kernel_size[0], ct.stride[0], ct.padding[0]
        Ci, Co = ct.in_channels, ct.out_channels
        if Ci != 64 or Co != 64 or (k, s_) not in ((3, 1), (5, 2), (3, 2)) or pad != s_ // 2:
            return False
        wfragT_ptr = self._pack(ct.weight.data, Ci, Co * k, k, 0, [Co] * k, list(range(k)))
        a = BConvtArgs()
        a.NP, a.N, a.Lin, a.Lout, a.K, a.S, a.pad, a.Ci, a.Co = NP, N, a_pw.L, Lo, k, s_, pad, Ci, Co
        a.dy, a.z = ptr16(dy), ptr16(z)
        a.ca, a.cb, a.cc = ptr(bn.ca), ptr(bn.cb), ptr(bn.cc)
        a.src, a.s_scale, a.s_shift, a.s_mean = ptr16(a_pw.t), ptr(a_pw.bn.scale), ptr(a_pw.bn.shift), ptr(a_pw.bn.mean)
        nparts = lib.trunet_bf16_convt_bwd_nparts()
        part = w.flat("ct16_partials", nparts * Ci * 2, zero=True)
        a.wfragT, a.dsrc, a.partials = wfragT_ptr, ptr16(dy_pw), ptr(part)
        a.prezero = 1 if w.take_clean("ct16_partials") else 0
        a.w_numel = self._wg_total
        a.w_partials, a.b_partials = self._wg_slot(ct.weight), self._wg_slot(ct.bias)
        a.b_stride, a.b_off = self._wg_total, 0
        if E.PROFILE is not None:
            by = 2 * (2 * Co * Lo + 2 * Ci * a_pw.L)          # dy, z rows once; source row in, its gradient out
            with _Timed("bconvt_bwd_kernel<%d, %d>" % (k, s_), float(by) * N, "L%d" % a_pw.L):
                rc = lib.trunet_bf16_convt_bwd(a, st)
        else:
            rc = lib.trunet_bf16_convt_bwd(a, st)
        if rc == L.TRUNET_ENOTSUP:
            w.pending.discard("ct16_partials")
            return False
        check(rc, "bf16_convt_bwd")
        self._bn_bwd(w, a_pw.bn, nparts, grads, part_name="ct16_partials")
        return True

    def _bwd_tr16(self, w, N, NP, ct, pw, a_pw, Lo, up, x1, x1_mask, skip, left, dy_x1, g_skip, dy_pw_name, grads, pw32=None):
        """FirstTrCNN / TrCNN / LastTrCNN (network.py:60-120): transposed conv, then the pointwise conv over [x1 | skip]"""
        dy, z, bn = up
        k, s_, pad = ct.kernel_size[0], ct.stride[0], ct.padding[0]
        Ci, Co = ct.in_channels, ct.out_channels
        dy_pw = self._get16(w, dy_pw_name, Ci, a_pw.L, NP)
        if pw32 is not None:
            # fp32 transposed conv (LAST_CT32): dy is the fp32 output cotangent [Co][Lo][NP]; weight / bias gradient and the
            # masked data gradient with its BatchNorm-backward sums on the fp32 kernels, the result rounded once to octets
            self._wgrad(w, N=N, NP=NP, P=Lo, M=Co, dz=dy, dz1=None, dz_L=Lo, dz_bn=None, W=ct.weight, ldw_m=k, ldw_c=Co * k,
                        segs=[pw32.seg(pos_off=pad - kk, woff=kk, pos_div=s_) for kk in range(k)], grads=grads, bias=ct.bias)
            dy_pw32 = w.get(dy_pw_name + ".f32", (Ci, a_pw.L, NP))
            segs = [TRUNetEngine._dz_seg(dy, None, None, Co, Lo, pos_mul=s_, pos_off=kk - pad, woff=kk) for kk in range(k)]
            nparts = self._gemm(w, N=N, NP=NP, P=a_pw.L, M=Ci, out=dy_pw32, out_L=a_pw.L, W=ct.weight.data, ldw_m=Co * k,
                                ldw_c=k, segs=segs, zmask=pw32.t, e0=a_pw.bn.scale, e1=a_pw.bn.shift, e2=a_pw.bn.mean,
                                stats=Ci)
            self._bn_bwd(w, a_pw.bn, nparts, grads)
            check(L.lib().trunet_bf16_from_frames_last(ptr(dy_pw32), ptr16(dy_pw), Ci, a_pw.L, NP, L.stream()),
                  "bf16_from_frames_last")
        elif not (FUSED_CONVT16 and bn is not None and self._convt_bwd16(w, N, NP, ct, a_pw, Lo, dy, z, bn, dy_pw, grads)):
            self._wgrad16(w, N=N, NP=NP, P=Lo, M=Co, dz=dy, dz1=z, dz_L=Lo, dz_bn=bn, W=ct.weight, ldw_m=k, ldw_c=Co * k,
                          segs=[a_pw.seg(pos_off=pad - kk, woff=kk, pos_div=s_) for kk in range(k)], grads=grads,
                          bias=ct.bias)
            segs = [self._dz_seg16(dy, z, bn, Co, Lo, pos_mul=s_, pos_off=kk - pad, woff=kk) for kk in range(k)]
            nparts = self._gemm16(w, N=N, NP=NP, P=a_pw.L, M=Ci, out=dy_pw, out_L=a_pw.L, W=ct.weight.data, ldw_m=Co * k,
                                  ldw_c=k, segs=segs, zmask=a_pw.t, e0=a_pw.bn.scale, e1=a_pw.bn.shift, e2=a_pw.bn.mean,
                                  stats=Ci)
            self._bn_bwd(w, a_pw.bn, nparts, grads)
        Lp = a_pw.L
        srcs = [x1.seg(pos_off=-left, woff=0)] + ([skip.seg(woff=x1.C)] if skip is not None else [])
        p0, p1 = max(0, left), min(Lp, x1.L + left)
        if p1 - p0 < x1.L:          # cropped positions of x1 (network.py:96-97 with a negative pad) get no gradient
            w.zero_crop(dy_x1, p0 - left, p1 - left)
        outs = [dict(out=dy_x1, src=x1_mask)] + ([dict(out=g_skip)] if skip is not None else [])
        self._pw_bwd16(w, N=N, NP=NP, P=Lp, M=pw.out_channels, dz=dy_pw, dz1=a_pw.t, dz_bn=a_pw.bn, W=pw.weight,
                       bias=pw.bias, segs=srcs, outs=outs, grads=grads)

    def _bwd_dsc16(self, w, N, NP, seq, a_pw, a_dw, up, prev, prev_mask, accum, dy_prev, dy_pw_name, grads):
        """DepthwiseSeparableConv1d (network.py:24-43): depthwise conv (fused dgrad + wgrad + BN-backward sums), then the
        pointwise conv; dy_prev receives the gradient of the block input (added to the skip gradient when ``accum``)."""
        lib, st = L.lib(), L.stream()
        dy, z, bn = up
        pw, dwc = seq[0], seq[3]
        k, s_ = dwc.kernel_size[0], dwc.stride[0]
        Cn = dwc.out_channels
        dy_pw = self._get16(w, dy_pw_name, Cn, a_pw.L, NP)
        nparts = lib.trunet_bf16_dw_nparts(NP, a_pw.L)
        part = w.flat("partials_dw", nparts * Cn * 2)
        wpart = w.flat("dw_w_partials", nparts * Cn * k)
        bpart = w.flat("dw_b_partials", nparts * Cn)
        args = (ptr16(dy), ptr16(z), ptr(bn.ca), ptr(bn.cb), ptr(bn.cc), ptr16(a_pw.t), ptr(a_pw.bn.scale),
                ptr(a_pw.bn.shift), ptr(a_pw.bn.mean), ptr(dwc.weight.data), ptr16(dy_pw), ptr(part), ptr(wpart), ptr(bpart),
                Cn, k, s_, a_pw.L, a_dw.L, NP, N, st)
        if E.PROFILE is not None:
            with _Timed("bdw_bwd_kernel<%d, %d>" % (k, s_), 2.0 * Cn * (2 * a_dw.L + 2 * a_pw.L) * N, "L%d" % a_pw.L):
                check(lib.trunet_bf16_dwconv_bwd(*args), "bf16_dwconv_bwd")
        else:
            check(lib.trunet_bf16_dwconv_bwd(*args), "bf16_dwconv_bwd")
        check(lib.trunet_reduce_partials(self._wg_slot(dwc.weight), ptr(wpart), nparts, Cn * k, 0, st), "reduce")
        check(lib.trunet_reduce_partials(self._wg_slot(dwc.bias), ptr(bpart), nparts, Cn, 0, st), "reduce")
        self._bn_bwd(w, a_pw.bn, nparts, grads, part_name="partials_dw")
        self._pw_bwd16(w, N=N, NP=NP, P=a_pw.L, M=pw.out_channels, dz=dy_pw, dz1=a_pw.t, dz_bn=a_pw.bn, W=pw.weight,
                       bias=pw.bias, segs=[prev.seg()], outs=[dict(out=dy_prev, src=prev_mask, accum=accum)], grads=grads)

    # ------------------------------------------------------------------ backward
    def backward(self, ctx, gout):
        acts, N, NP, w, gen = ctx
        self._check_gen(w, gen)
        net = self.net
        lib, st = L.lib(), L.stream()
        grads = {}
        self._wg_begin(w)
        gout = gout.contiguous()
        last = acts["dec5"]
        if isinstance(last, Act16):
            dyt = self._get16(w, "dy:dec5", last.C, last.L, NP)
            check(lib.trunet_bf16_from_ncl(ptr(gout), ptr16(dyt), N, last.C, last.L, NP, st), "bf16_from_ncl")
        else:       # LAST_CT32: the last transposed conv ran in fp32, its cotangent stays fp32 frames-last
            dyt = w.get("dy:dec5.f32", (last.C, last.L, NP))
            check(lib.trunet_to_frames_last(ptr(gout), ptr(dyt), N, last.C, last.L, NP, st), "to_frames_last")
        up = (dyt, last.t, None)
        for i in range(5, -1, -1):
            seq = (net.decoder[i].LastTrCNN if i == 5 else net.decoder[i].TrCNN) if i > 0 else net.decoder[0].FirstTrCNN
            if i > 0:
                x1, x1n = acts["dec%d" % (i - 1)], "dec%d" % (i - 1)
                skip = acts["enc%d" % (5 - i)]
                left = (skip.L - x1.L) // 2
                g_skip = self._get16(w, "dy:enc%d" % (5 - i), skip.C, skip.L, NP)
            else:
                x1n = "fgru"
                x1, skip, left, g_skip = acts[x1n], None, 0, None
            dy_x1 = self._get16(w, "dy:" + x1n, x1.C, x1.L, NP)
            self._bwd_tr16(w, N, NP, seq[3], seq[0], acts["dec%d.pw" % i], acts["dec%d" % i].L, up, x1, x1, skip, left,
                           dy_x1, g_skip, "dy:dec%d.pw" % i, grads, pw32=(acts.get("dec5.pw32") if i == 5 else None))
            up = (dy_x1, x1.t, x1.bn)

        # -------- FGRU (fp32)
        enc5 = acts["enc5"]
        if "enc5.f32" in acts:
            fg, enc5f = acts["fgru.f32"], acts["enc5.f32"]
            dyf = self._to32(w, "dy:fgru.f32", up[0], fg.C, fg.L, NP)
            dy5f = w.get("dy:enc5.f32", (enc5f.C, enc5f.L, NP))
            self._bwd_fgru(w, N, NP, net.FGRU, (dyf, fg.t, fg.bn), acts["hout"], enc5f, enc5f, dy5f, grads)
            dy5 = self._to16(w, "dy:enc5", dy5f, enc5.C, enc5.L, NP)
        else:
            dy5 = self._get16(w, "dy:enc5", enc5.C, enc5.L, NP)
            self._bwd_fgru16(w, N, NP, net.FGRU, up, acts["hout"], acts["hout16"], enc5, dy5, grads)
        up = (dy5, enc5.t, enc5.bn)

        for i in range(5, 0, -1):
            prev = acts["enc%d" % (i - 1)]
            dy_prev = self._get16(w, "dy:enc%d" % (i - 1), prev.C, prev.L, NP)
            self._bwd_dsc16(w, N, NP, net.encoder[i].DepthwiseSeparableConv1d, acts["enc%d.pw" % i], acts["enc%d" % i], up,
                            prev, prev, True, dy_prev, "dy:enc%d.pw" % i, grads)
            up = (dy_prev, prev.t, prev.bn)

        # -------- first conv (its output is ReLU-only: up[0] is already masked)
        c0 = net.encoder[0].StandardConv1d[0]
        xa = acts["x"]
        k, s_, pad = c0.kernel_size[0], c0.stride[0], c0.padding[0]
        self._wgrad16(w, N=N, NP=NP, P=acts["enc0"].L, M=c0.out_channels, dz=up[0], dz_L=acts["enc0"].L, dz_bn=None,
                      W=c0.weight, ldw_m=xa.C * k, ldw_c=k,
                      segs=[xa.seg(pos_mul=s_, pos_off=kk - pad, woff=kk) for kk in range(k)], grads=grads, bias=c0.bias)
        self._wg_finish(grads)
        return grads
