"""How far apart are the gradients of the full configs[1] step under perturbations of fp32-rounding size?
  base: fp32-MFMA kernels;  pert: the same kernels, noisy audio multiplied by (1 + 2e-7 randn);  split: bf16-split kernels."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import network_ref as nr, weights as W
from tinyrecurrentunet_amd import _lib, network as hn, stft_loss as sl, util
lib = _lib.lib()
CFG = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200], sc_lambda=0.5, mag_lambda=0.5, band="full")
B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 64), 64000
clean, noisy = W.synth_pairs(B, L, seed=1234)
ref = W.fill_state_dict(nr.TRUNet(input_size=4), seed=0)
net = hn.TRUNet(input_size=4)
net.load_state_dict(ref.state_dict())
net.cuda().train()
mr = sl.MultiResolutionSTFTLoss(**CFG).cuda()
cg, ng = clean.cuda(), noisy.cuda()
g = torch.Generator(device="cuda").manual_seed(1)
ng2 = ng * (1.0 + 2e-7 * torch.randn(ng.shape, generator=g, device="cuda"))
print("relative size of the input perturbation: %.2e" % ((ng2 - ng).norm() / ng.norm()).item())
def run(x3, nz):
    lib.trunet_gemm_x3_enable(x3)
    net.load_state_dict(ref.state_dict())
    net.zero_grad()
    loss, info = util.loss_fn(net, (cg, nz), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
base, pert, split, split_pert = run(0, ng), run(0, ng2), run(1, ng), run(1, ng2)
def dist(a, b):
    es = []
    for n, x in a[1].items():
        y = b[1][n]
        if y.abs().max().item() < 1e-3:
            continue
        es.append(((x - y).norm() / y.norm()).item())
    return "median %.2e max %.2e" % (np.median(es), max(es))
print("B = %d  losses: %.7f %.7f %.7f %.7f" % (B, base[0], pert[0], split[0], split_pert[0]))
print("  fp32-MFMA vs fp32-MFMA on the perturbed input : " + dist(pert, base))
print("  split     vs fp32-MFMA                        : " + dist(split, base))
print("  split     vs split on the perturbed input     : " + dist(split_pert, split))
lib.trunet_gemm_x3_enable(1)
