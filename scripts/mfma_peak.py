import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyrecurrentunet_amd import _lib as L
lib = L.lib()
for blocks in (256, 512):
    out = torch.empty(blocks * 256, device="cuda")
    iters = 20000
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.time()
        L.check(lib.trunet_debug_mfma_peak(out.data_ptr(), blocks, iters, L.stream()))
        torch.cuda.synchronize(); dt = time.time() - t0
        fl = blocks * 4 * 16 * iters * 4096.0
        print("blocks %d: %.2f ms, %.1f TFLOP/s fp32 MFMA (=> clock %.2f GHz if 256 flop/clk/CU)" % (blocks, dt * 1e3, fl / dt / 1e12, fl / dt / 256 / 256 / 1e9 / min(blocks / 256, 1)))
