"""Diagnostic: where one frame's cycles go inside stream_fwd_kernel (workgroup 0, first frame), from s_memtime stamps of a
-DSF_STAMPS build (scripts/dbg/libsf_stamps.so, built by `bash scripts/build_dbg.sh`).  Prints cycles per phase."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tinyrecurrentunet_amd import export, network as hn  # noqa: E402

lib = C.CDLL(os.path.join(ROOT, "scripts", "dbg", os.environ.get("SF_LIB", "libsf_stamps.so")))
lib.trunet_stream_fwd.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_int, C.c_int64, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.trunet_stream_fwd_scratch_floats.restype = C.c_size_t
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
torch.manual_seed(0)
net = hn.TRUNet(input_size=4).cuda().eval()
f = export.FoldedTRUNet.from_module(net)
NO, BN = len(f.offsets), f.blob.numel()


def launch():
    return lib.trunet_stream_fwd(x.data_ptr(), y.data_ptr(), f.blob.data_ptr(), f._offs, NO, BN, scratch.data_ptr(), None, None, N,
                                 4, st)

x = torch.randn(N, 4, 257, device="cuda")
y = torch.empty(N, 8, 257, device="cuda")
grid = lib.trunet_stream_fwd_grid(N)
nf = lib.trunet_stream_fwd_scratch_floats(grid)
scratch = torch.zeros(nf, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(3):
    assert launch() == 0
torch.cuda.synchronize()
t0 = time.time()
for _ in range(20):
    launch()
torch.cuda.synchronize()
dt = (time.time() - t0) / 20
s = scratch[grid * 45056:].view(torch.int64)[:64].cpu().numpy().astype(np.int64)
names = {0: "x+first conv", 1: "enc1", 2: "enc2-5 + GRU proj", 7: "W_hh load", 23: "GRU recurrence", 26: "guards",
         8: "fgru pw + dec0", 9: "dec1-4", 13: "dec5"}
order = [0, 1, 2, 7, 23, 26, 8, 9, 13, 14]
tot = s[14] - s[0]
print("N=%d grid=%d: %.3f ms per launch; frame 0 of workgroup 0: %d cycles" % (N, grid, dt * 1e3, tot))
for a, b in zip(order[:-1], order[1:]):
    print("  %-20s %8d  %5.1f %%" % (names[a], s[b] - s[a], 100.0 * (s[b] - s[a]) / tot))
enc = [2, 3, 4, 5, 6, 10, 11, 12]
print("  encoder.2..5, projection x3 (pointwise + depthwise + save each): " + " ".join(str(int(s[b] - s[a])) for a, b in zip(enc[:-1], enc[1:])))
for i in range(1, 5):
    a = 9 if i == 1 else 12 + 2 * i
    print("  dec%d: pw %d  convT+restore %d" % (i, s[13 + 2 * i] - s[a], s[14 + 2 * i] - s[13 + 2 * i]))
print("  enc3 (it=1, 2 column tiles = 128 MFMAs/wave): wait for own fragments %d  request next %d  pw %d  guards+dw+syncs %d" % (
    s[28] - s[27], s[29] - s[28], s[30] - s[29], s[31] - s[30]))
d = s[32:32 + 17]
print("  dec4 pw (wave 0): entry->group0 %d; per group [mm1, mm2, stores, to next]: %s" % (
    d[1] - d[0], " ".join("[%d %d %d]" % (d[2 + 4 * g] - d[1 + 4 * g], d[3 + 4 * g] - d[2 + 4 * g], d[4 + 4 * g] - d[3 + 4 * g]) for g in range(4))))
