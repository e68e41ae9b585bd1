"""The full configs[1] train step (64 x 4 s) with the bf16-split kernels on and off: how far apart are the gradients of two
fp32-accurate implementations of the same step?  (Both are compared with the fp32 oracle by tests/test_configs_gpu.py.)"""
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
res = {}
for mode in (0, 1):
    lib.trunet_gemm_x3_enable(mode)
    net.load_state_dict(ref.state_dict())
    net.zero_grad()
    loss, info = util.loss_fn(net, (cg, ng), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
    loss.backward()
    torch.cuda.synchronize()
    res[mode] = (float(loss), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
print("loss: fp32-MFMA %.7f  split %.7f" % (res[0][0], res[1][0]))
es = []
for n, a in res[1][1].items():
    b = res[0][1][n]
    if b.abs().max().item() < 1e-3:
        continue
    es.append(((a - b).norm() / b.norm()).item())
print("B = %d: gradients of the two paths against each other, %d tensors: relative L2 median %.2e  max %.2e" % (B, len(es), np.median(es), max(es)))
