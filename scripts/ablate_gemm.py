"""Diagnostic: time conv_gemm with parts of the per-chunk bookkeeping disabled (results are wrong on purpose)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyrecurrentunet_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dbg", "libtrunet_hip_abl.so")
from tinyrecurrentunet_amd._lib import GemmArgs, make_seg, ptr, check, PRO_BNRELU, EPI_BIAS, EPI_STATS
lib = L.lib()

def run(N, Ln, K, M, abl):
    NP = (N + 127) // 128 * 128
    x = torch.randn(K, Ln, NP, device="cuda"); out = torch.empty(M, Ln, NP, device="cuda")
    W = torch.randn(M, K, device="cuda") * 0.05
    b = torch.zeros(M, device="cuda"); s = torch.ones(K, device="cuda"); t = torch.zeros(K, device="cuda")
    part = torch.empty(2048 * M * 2, device="cuda")
    a = GemmArgs()
    a.NP, a.N, a.P, a.p_begin = NP, N, Ln, 0
    a.M, a.m_out_off, a.out_L, a.out_pos_off = M, 0, Ln, 0
    a.ldw_m, a.ldw_c, a.w_m_off, a.nseg = K, 1, 0, 1
    a.seg[0] = make_seg(x, K, Ln, mode=PRO_BNRELU, c0=s, c1=t)
    a.out, a.W, a.bias, a.partials, a.M_stat = ptr(out), ptr(W), ptr(b), ptr(part), M
    a.epi = EPI_BIAS | (0 if abl & 2048 else EPI_STATS) | (abl & ~2048)
    for _ in range(2):
        check(lib.trunet_conv_gemm(a, L.stream()))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        check(lib.trunet_conv_gemm(a, L.stream()))
    torch.cuda.synchronize(); dt = (time.time() - t0) / 5
    print("K=%d M=%d abl=%5d: %7.3f ms  %6.1f TF" % (K, M, abl, dt * 1e3, 2.0 * N * Ln * M * K / dt / 1e12), flush=True)

for K, M in ((128, 128), (192, 64)):
    for abl in (0, 256, 2048, 256 | 2048, 4096, 1024, 512, 512 | 1024, 512 | 1024 | 4096):
        run(32064, 128, K, M, abl)
