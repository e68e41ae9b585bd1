"""Micro-benchmark of trunet_pw_bwd (fused pointwise-conv backward) at the train-step shapes, through the C ABI."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyrecurrentunet_amd import _lib as L
if os.environ.get("TRUNET_HIP_LIB"):      # diagnostic builds (ablations) of the library
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["TRUNET_HIP_LIB"])
from tinyrecurrentunet_amd._lib import (DG_ACCUM, DG_MASK, DG_STATS, DG_STORE, PRO_BNBWD, PRO_BNRELU, PRO_NONE, PwBwdArgs,
                                        check, make_seg, ptr)


def run(N, P, M, srcs, reps=3, label=""):
    """srcs: list of (nchan, kind) kind in bn_accum / bn / raw"""
    dev = "cuda"
    NP = (N + 255) // 256 * 256
    K = sum(c for c, _ in srcs)
    dy, z = torch.randn(M, P, NP, device=dev), torch.randn(M, P, NP, device=dev)
    ca, cb, cc = torch.ones(M, device=dev), torch.zeros(M, device=dev) + 0.1, torch.zeros(M, device=dev)
    W = torch.randn(M, K, 1, device=dev) * 0.1
    lib = L.lib()
    f = PwBwdArgs()
    fw = f.w
    fw.NP, fw.N, fw.P, fw.p_begin = NP, N, P, 0
    fw.M, fw.a_L, fw.a_pos_off, fw.a_m_off = M, P, 0, 0
    fw.ldw_m, fw.ldw_c, fw.w_m_off, fw.nseg, fw.w_numel = K, 1, 0, len(srcs), W.numel()
    fw.a0, fw.a1, fw.a_mode = ptr(dy), ptr(z), PRO_BNBWD
    fw.ac0, fw.ac1, fw.ac2 = ptr(ca), ptr(cb), ptr(cc)
    npw = lib.trunet_conv_wgrad_nparts()
    wp = torch.empty(npw * W.numel(), device=dev)
    bp = torch.empty(npw * M, device=dev)
    fw.w_partials, fw.b_partials, fw.b_stride, fw.b_off = ptr(wp), ptr(bp), M, 0
    f.W = ptr(W)
    nparts = lib.trunet_pw_bwd_nparts()
    keep, woff, rows = [], 0, 2 * M
    for i, (C, kind) in enumerate(srcs):
        zs = torch.randn(C, P, NP, device=dev)
        sc, sh, mean = torch.ones(C, device=dev), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
        out = torch.zeros(C, P, NP, device=dev)
        part = torch.empty(nparts * C * 2, device=dev)
        fw.seg[i] = make_seg(zs, C, P, woff=woff, mode=PRO_BNRELU if kind.startswith("bn") else PRO_NONE, c0=sc, c1=sh)
        d = f.dg[i]
        d.out = ptr(out)
        fl = DG_STORE
        rows += C + C           # source in, gradient out
        if kind != "raw":
            fl |= DG_MASK | DG_STATS
            d.zmask, d.e2, d.partials = ptr(zs), ptr(mean), ptr(part)
            if "accum" in kind:
                fl |= DG_ACCUM
                rows += C
        d.flags = fl
        keep.append((zs, sc, sh, mean, out, part))
        woff += C
    for _ in range(2):
        check(lib.trunet_pw_bwd(f, L.stream()))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps):
        check(lib.trunet_pw_bwd(f, L.stream()))
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    fl_ = 4.0 * N * P * M * K
    by = 4.0 * N * P * rows
    print("pw_bwd %-10s N=%6d P=%3d M=%3d K=%s: %7.3f ms  %6.1f TF  %5.2f TB/s (algorithmic)" % (
        label, N, P, M, "+".join(str(c) for c, _ in srcs), dt * 1e3, fl_ / dt / 1e12, by / dt / 1e12), flush=True)


if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    if which in ("all", "enc"):
        run(32064, 128, 128, [(128, "bn_accum")], reps, "enc")
    if which in ("all", "enc1"):
        run(32064, 128, 128, [(64, "bn_accum")], reps, "enc1")
    if which in ("all", "dec"):
        run(32064, 128, 64, [(64, "bn"), (128, "raw")], reps, "dec")
    if which in ("all", "fgru"):
        run(32064, 16, 64, [(128, "raw")], reps, "fgru")
