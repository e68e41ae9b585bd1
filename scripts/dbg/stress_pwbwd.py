"""Diagnostic: call trunet_pw_bwd many times on identical operands and compare every output bit for bit."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinyrecurrentunet_amd import _lib as L
from tinyrecurrentunet_amd._lib import (DG_ACCUM, DG_MASK, DG_STATS, DG_STORE, PRO_BNBWD, PRO_BNRELU, PRO_NONE, PwBwdArgs,
                                        check, make_seg, ptr)


def run(N, P, M, srcs, reps, label, noise=False):
    dev = "cuda"
    g = torch.Generator(device=dev); g.manual_seed(1)
    rnd = lambda *s: torch.randn(*s, generator=g, device=dev)
    NP = (N + 255) // 256 * 256
    K = sum(c for c, _ in srcs)
    dy, z = rnd(M, P, NP), rnd(M, P, NP)
    ca, cb, cc = rnd(M) * 0.5 + 1, rnd(M) * 0.1, rnd(M) * 0.01
    W = rnd(M, K, 1) * 0.1
    lib = L.lib()
    f = PwBwdArgs(); fw = f.w
    fw.NP, fw.N, fw.P, fw.p_begin = NP, N, P, 0
    fw.M, fw.a_L, fw.a_pos_off, fw.a_m_off = M, P, 0, 0
    fw.ldw_m, fw.ldw_c, fw.w_m_off, fw.nseg, fw.w_numel = K, 1, 0, len(srcs), W.numel()
    fw.a0, fw.a1, fw.a_mode = ptr(dy), ptr(z), PRO_BNBWD
    fw.ac0, fw.ac1, fw.ac2 = ptr(ca), ptr(cb), ptr(cc)
    npw = lib.trunet_conv_wgrad_nparts()
    wp = torch.empty(npw * W.numel(), device=dev); bp = torch.empty(npw * M, device=dev)
    fw.w_partials, fw.b_partials, fw.b_stride, fw.b_off = ptr(wp), ptr(bp), M, 0
    f.W = ptr(W)
    nparts = lib.trunet_pw_bwd_nparts()
    keep, outs, prevs, parts, woff = [], [], [], [], 0
    for i, (C, kind) in enumerate(srcs):
        zs = rnd(C, P, NP)
        sc, sh, mean = rnd(C) * 0.3 + 1, rnd(C) * 0.2, rnd(C) * 0.1
        out = torch.zeros(C, P, NP, device=dev)
        prev = rnd(C, P, NP) if "accum" in kind else None
        part = torch.empty(nparts * C * 2, device=dev)
        fw.seg[i] = make_seg(zs, C, P, woff=woff, mode=PRO_BNRELU if kind.startswith("bn") else PRO_NONE, c0=sc, c1=sh)
        d = f.dg[i]; d.out = ptr(out)
        fl = DG_STORE
        if kind != "raw":
            fl |= DG_MASK | DG_STATS
            d.zmask, d.e2, d.partials = ptr(zs), ptr(mean), ptr(part)
            if "accum" in kind: fl |= DG_ACCUM
        d.flags = fl
        keep.append((zs, sc, sh, mean)); outs.append(out); prevs.append(prev); parts.append(part); woff += C
    junk = torch.empty(64 << 20, device=dev)
    ref, bad = None, 0
    for it in range(reps):
        for o, pv in zip(outs, prevs):
            if pv is not None: o.copy_(pv)
        if noise and it % 3 == 0:
            junk.normal_()                      # evict caches / change timing
        if noise and it % 5 == 0:
            torch.cuda.synchronize()
        check(lib.trunet_pw_bwd(f, L.stream()))
        cur = [wp.clone(), bp.clone()] + [o.clone() for o in outs] + [p.clone() for p in parts]
        if ref is None:
            ref = cur; continue
        for k, (a, b) in enumerate(zip(cur, ref)):
            if not torch.equal(a, b):
                bad += 1
                d = (a - b).abs()
                idx = int(d.reshape(-1).argmax())
                print("%s it %d tensor %d differs: max %.3g at flat index %d of %d, nnz %d" % (label, it, k, float(d.max()), idx, a.numel(), int((d > 0).sum())), flush=True)
                break
        if bad > 5: break
    print("%s: %d bad of %d" % (label, bad, reps), flush=True)


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    run(300, 128, 128, [(128, "bn_accum")], reps, "enc N300 P128", noise=True)
    run(300, 64, 128, [(128, "bn_accum")], reps, "enc N300 P64", noise=True)
    run(300, 128, 128, [(64, "bn_accum")], reps, "enc1 N300", noise=True)
    run(300, 128, 64, [(64, "bn"), (128, "raw")], reps, "dec N300", noise=True)
    run(2000, 32, 128, [(128, "bn_accum")], reps // 3, "enc N2000 P32", noise=True)
