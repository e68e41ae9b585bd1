"""Micro-benchmark of conv_gemm launches through the C ABI (pw conv, BN+ReLU prologue, stats)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyrecurrentunet_amd import _lib as L
if os.environ.get("TRUNET_HIP_LIB"):      # diagnostic builds of the library
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["TRUNET_HIP_LIB"])
from tinyrecurrentunet_amd._lib import GemmArgs, make_seg, ptr, check, PRO_BNRELU, PRO_NONE, EPI_BIAS, EPI_STATS

def run(N, Ln, K, M, mode, reps=5):
    NP = (N + 255) // 256 * 256
    dev = "cuda"
    x = torch.randn(K, Ln, NP, device=dev)
    out = torch.empty(M, Ln, NP, device=dev)
    W = torch.randn(M, K, device=dev) * 0.05
    b = torch.zeros(M, device=dev); s = torch.ones(K, device=dev); t = torch.zeros(K, device=dev)
    part = torch.empty(2048 * M * 2, device=dev)
    a = GemmArgs()
    a.NP, a.N, a.P, a.p_begin = NP, N, Ln, 0
    a.M, a.m_out_off, a.out_L, a.out_pos_off = M, 0, Ln, 0
    a.ldw_m, a.ldw_c, a.w_m_off, a.nseg = K, 1, 0, 1
    a.seg[0] = make_seg(x, K, Ln, mode=mode, c0=s, c1=t)
    a.out, a.W, a.bias, a.partials, a.M_stat = ptr(out), ptr(W), ptr(b), ptr(part), M
    a.epi = EPI_BIAS | EPI_STATS
    lib = L.lib()
    for _ in range(2):
        check(lib.trunet_conv_gemm(a, L.stream()))
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(reps):
        check(lib.trunet_conv_gemm(a, L.stream()))
    torch.cuda.synchronize(); dt = (time.time() - t0) / reps
    fl = 2.0 * N * Ln * M * K
    by = 4.0 * N * Ln * (M + K)
    print("gemm N=%6d L=%3d K=%3d M=%3d mode=%d: %7.3f ms  %6.1f TF  %6.2f TB/s" % (N, Ln, K, M, mode, dt * 1e3, fl / dt / 1e12, by / dt / 1e12), flush=True)

if __name__ == "__main__":
    for N in (2048, 32064):
        run(N, 128, 128, 128, PRO_BNRELU)
    run(32064, 128, 128, 128, PRO_NONE)
    run(32064, 16, 128, 128, PRO_BNRELU)
    run(32064, 128, 64, 128, PRO_BNRELU)
    run(32064, 128, 192, 64, PRO_BNRELU)
    run(32064, 128, 128, 8, PRO_BNRELU)
