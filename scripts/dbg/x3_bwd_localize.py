"""Which fused backward kernel moves the full-size gradients when it multiplies on the bf16 MFMA?  Same forward (fp32 MFMA) in
every mode; modes: 0 all fp32 MFMA, 4 pw_bwd split only, 8 convt_bwd split only, 2 both."""
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
for tag, mode in (("fp32", 0), ("fp32 again", 0), ("pw_bwd", 4), ("convt_bwd", 8), ("both", 2), ("both again", 2)):
    lib.trunet_gemm_x3_enable(mode)
    net.load_state_dict(ref.state_dict())
    net.zero_grad()
    loss, info = util.loss_fn(net, (cg, ng), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
    loss.backward()
    torch.cuda.synchronize()
    res[tag] = (float(loss), {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None})
lib.trunet_gemm_x3_enable(2)
base = res["fp32"][1]
for tag in list(res)[1:]:
    es = []
    for n, a in res[tag][1].items():
        b = base[n]
        if b.abs().max().item() < 1e-3:
            continue
        es.append((((a - b).norm() / b.norm()).item(), n))
    es.sort()
    print("B = %d %-11s loss %.7f vs fp32: relative L2 median %.2e  max %.2e (%s), worst five: %s" % (
        B, tag, res[tag][0], np.median([e for e, _ in es]), es[-1][0], es[-1][1], ", ".join("%s %.1e" % (n, e) for e, n in es[-5:])), flush=True)
