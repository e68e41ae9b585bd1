"""B = 64: is the gradient of the loss with respect to the net output (mask + iSTFT + L1 + MR-STFT) hypersensitive to
perturbations of the size by which the two fp32 paths differ at the output (2e-6)?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import network_ref as nr, weights as W
from tinyrecurrentunet_amd import _lib, dataset as ds, network as hn, stft_loss as sl, util
CFG = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200], sc_lambda=0.5, mag_lambda=0.5, band="full")
B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 64000
clean, noisy = W.synth_pairs(B, L, seed=1234)
ref = W.fill_state_dict(nr.TRUNet(input_size=4), seed=0)
net = hn.TRUNet(input_size=4)
net.load_state_dict(ref.state_dict())
net.cuda().train()
mr = sl.MultiResolutionSTFTLoss(**CFG).cuda()
cg, ng = clean.cuda(), noisy.cuda()
feats = ds.stft_features(ng[:, 0].contiguous(), pcen=True)
T = feats.shape[0] // B
plan = util._fused_loss_plan(mr, 1.0, feats.device)
def gnet(y):
    y = y.detach().clone().requires_grad_(True)
    loss, vals = util._FusedLossFn.apply(y, cg[:, 0].contiguous(), T, 0.5, 1.0, plan, 0.5, 0.5)
    loss.backward()
    return y.grad
ys = {}
for kind in ("fp32", "bf16x3"):
    _lib.set_fp32_mfma(kind)
    net.load_state_dict(ref.state_dict())
    with torch.no_grad():
        ys[kind] = net(feats)
_lib.set_fp32_mfma("fp32")
y0, y1 = ys["fp32"], ys["bf16x3"]
d = y1 - y0
print("net output, split vs fp32-MFMA: relative L2 %.2e, max|d| %.2e" % ((d.norm() / y0.norm()).item(), d.abs().max().item()))
g0, g1 = gnet(y0), gnet(y1)
gen = torch.Generator(device="cuda").manual_seed(3)
noise = torch.randn(y0.shape, generator=gen, device="cuda") * d.std()
g2 = gnet(y0 + noise)
g3 = gnet(y0 + d[torch.randperm(d.shape[0], device="cuda", generator=gen)])      # the same differences, frames shuffled
rel = lambda a, b: ((a - b).norm() / b.norm()).item()
print("loss gradient w.r.t. the net output: |g| %.3e, max %.3e" % (g0.norm().item(), g0.abs().max().item()))
print("  split output vs fp32-MFMA output         : %.2e" % rel(g1, g0))
print("  fp32-MFMA output + white noise (same rms): %.2e" % rel(g2, g0))
print("  fp32-MFMA output + the differences, frames shuffled: %.2e" % rel(g3, g0))
# where does the difference of the gradient live?
e = (g1 - g0).reshape(-1).abs()
top = torch.topk(e, 10)
tot = (e.double() ** 2).sum().item()
print("  share of the squared gradient difference in its 10 / 1000 largest elements: %.3f / %.3f" % (
    (top.values.double() ** 2).sum().item() / tot, (torch.topk(e, 1000).values.double() ** 2).sum().item() / tot))
for i in top.indices[:5].tolist():
    n, rem = divmod(i, 8 * 257)
    ch, k = divmod(rem, 257)
    print("    frame %d ch %d bin %d: g %.3e vs %.3e; y (ch 2,3,6,7) fp32 %s split %s" % (
        n, ch, k, g1.reshape(-1)[i].item(), g0.reshape(-1)[i].item(),
        [round(y0[n, c, k].item(), 7) for c in (2, 3, 6, 7)], [round(y1[n, c, k].item(), 7) for c in (2, 3, 6, 7)]))
