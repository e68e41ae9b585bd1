"""Time the HIP TRU-Net body (fwd + bwd) at a given frame count; prints per-phase ms."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tinyrecurrentunet_amd import network as hn  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32064
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
torch.manual_seed(0)
net = hn.TRUNet(input_size=4).cuda().train()
x = torch.randn(N, 4, 257, device="cuda")
cot = torch.randn(N, 8, 257, device="cuda")
for it in range(2 + steps):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.time(); tf = 0.0
    torch.cuda.synchronize(); a = time.time()
    y = net(x)
    torch.cuda.synchronize(); b = time.time()
    y.backward(cot)
    torch.cuda.synchronize(); c = time.time()
    for p in net.parameters():
        p.grad = None
    if it >= 2:
        print("step %d fwd %.2f ms bwd %.2f ms" % (it, (b - a) * 1e3, (c - b) * 1e3), flush=True)
torch.cuda.synchronize()
dt = (time.time() - t0) / steps
print("N=%d  %.2f ms/step  %.0f frames/s  (%.1f TFLOP/s of 94.09 MFLOP/frame)" % (N, dt * 1e3, N / dt, N / dt * 94.09e6 / 1e12))
print("mem GB", torch.cuda.max_memory_allocated() / 2**30)
