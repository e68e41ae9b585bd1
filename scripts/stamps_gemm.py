"""Diagnostic: run conv_gemm launches on the stamp-instrumented library and print where the cycles go."""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyrecurrentunet_amd import _lib as L
L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts", "dbg", "libtrunet_hip_stamps.so")
from tinyrecurrentunet_amd._lib import GemmArgs, make_seg, ptr, check, PRO_BNRELU, PRO_NONE, EPI_BIAS, EPI_STATS
lib = L.lib()
lib.trunet_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
lib.trunet_debug_set_stamp_buffer.restype = ctypes.c_int
stamps = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
check(lib.trunet_debug_set_stamp_buffer(stamps.data_ptr()))

def run(N, Ln, K, M, mode):
    NP = (N + 127) // 128 * 128
    x = torch.randn(K, Ln, NP, device="cuda"); out = torch.empty(M, Ln, NP, device="cuda")
    W = torch.randn(M, K, device="cuda") * 0.05
    b = torch.zeros(M, device="cuda"); s = torch.ones(K, device="cuda"); t = torch.zeros(K, device="cuda")
    part = torch.empty(2048 * M * 2, device="cuda")
    a = GemmArgs()
    a.NP, a.N, a.P, a.p_begin = NP, N, Ln, 0
    a.M, a.m_out_off, a.out_L, a.out_pos_off = M, 0, Ln, 0
    a.ldw_m, a.ldw_c, a.w_m_off, a.nseg = K, 1, 0, 1
    a.seg[0] = make_seg(x, K, Ln, mode=mode, c0=s, c1=t)
    a.out, a.W, a.bias, a.partials, a.M_stat = ptr(out), ptr(W), ptr(b), ptr(part), M
    a.epi = EPI_BIAS | EPI_STATS
    for _ in range(3):
        check(lib.trunet_conv_gemm(a, L.stream()))
    torch.cuda.synchronize(); t0 = time.time()
    check(lib.trunet_conv_gemm(a, L.stream()))
    torch.cuda.synchronize(); dt = time.time() - t0
    st = stamps.view(512, 16).cpu().double()
    tot, rt, tX, tY, tB, tYl, steps, nph = [st[:, i].mean().item() for i in range(8)]
    print("   wait-DMA %.0f/step  transform %.0f/step  wait+transform+epilogue %.0f/step  dma-issue+iter %.0f/step" % tuple(st[:, i].mean().item() / steps for i in (8, 9, 10, 11)))
    print("K=%d M=%d L=%d: %.3f ms %.1f TF | per half: cycles %.0f  realtime %.1f us => clock %.2f GHz | steps %.0f | X %.0f/step  Y %.0f/step (last-chunk Y %.0f tot)  barrier-wait %.0f/phase" % (
        K, M, Ln, dt * 1e3, 2.0 * N * Ln * M * K / dt / 1e12, tot, rt / 100.0, tot / (rt * 10.0), steps, tX / steps, tY / steps, tYl, tB / nph))

run(32064, 128, 128, 128, PRO_BNRELU)
run(32064, 128, 192, 64, PRO_BNRELU)
run(32064, 128, 64, 128, PRO_BNRELU)
