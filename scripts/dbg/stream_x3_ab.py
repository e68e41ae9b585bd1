"""A/B of the single-launch eval forward: trunet_stream_fwd (fp32 MFMA) vs trunet_stream_fwd_x3 (bf16 MFMA through the three-term
split in the layers stream_fwd_x3.hip names): agreement, error of both against the float64 oracle, time per 1024 frames."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import network_ref as nr, weights as W
from tinyrecurrentunet_amd import _lib as L
if os.environ.get("TRUNET_HIP_LIB"):      # A/B builds of the library (other SFX_MASK values)
    L.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), os.environ["TRUNET_HIP_LIB"])
from tinyrecurrentunet_amd import export, network as hn
print("SFX_MASK", L.lib().trunet_stream_fwd_x3_mask())

torch.manual_seed(0)
ref = W.fill_state_dict(nr.TRUNet(input_size=4), seed=2)
net = hn.TRUNet(input_size=4)
net.load_state_dict(ref.state_dict())
net.cuda().eval()
refd = W.fill_state_dict(nr.TRUNet(input_size=4), seed=2).double().eval()
f = export.FoldedTRUNet.from_module(net, tgru=True)
for N in (1024, 255, 1):
    g = torch.Generator().manual_seed(4 + N)
    x = torch.randn(N, 4, 257, generator=g)
    xg = x.cuda()
    with torch.no_grad():
        yo = refd(x.double()) if N <= 255 else None
    y0 = f.use_x3(False)(xg).cpu()
    y1 = f.use_x3(True)(xg).cpu()
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    print("N = %4d: split vs fp32-MFMA kernel %.2e" % (N, rel(y1, y0)), ("; vs float64: fp32-MFMA %.2e, split %.2e" % (rel(y0, yo), rel(y1, yo))) if yo is not None else "", flush=True)
# stateful step
h0, h1 = f.new_state(64), f.new_state(64)
xs = torch.randn(64, 4, 257, device="cuda")
for t in range(3):
    a = f.use_x3(False).stream_step(xs, h0)
    b = f.use_x3(True).stream_step(xs, h1)
print("stateful, 3 steps: output %.2e, state %.2e" % (float((a - b).abs().max() / a.abs().max()), float((h0 - h1).abs().max() / h0.abs().max())))
xg = torch.randn(1024, 4, 257, device="cuda")
for kind in (False, True, False, True):
    f.use_x3(kind)
    for _ in range(20):
        f(xg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        f(xg)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    print("%-9s %.4f ms per 1024 frames = %.0f x real time" % ("split" if kind else "fp32-MFMA", ms, 1024 * 0.008 / (ms * 1e-3)), flush=True)
