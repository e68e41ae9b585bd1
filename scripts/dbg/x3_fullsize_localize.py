"""Where do the two paths of the full configs[1] step part?  Every workspace tensor (saved activations, BatchNorm states,
activation gradients) of the step with the bf16-split kernels on against the fp32-MFMA kernels."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import network_ref as nr, weights as W
from tinyrecurrentunet_amd import _lib, network as hn, stft_loss as sl, util
from tinyrecurrentunet_amd.engine import BNState
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
snaps = {}
for mode in (0, 1):
    lib.trunet_gemm_x3_enable(mode)
    net.load_state_dict(ref.state_dict())
    net.zero_grad()
    loss, info = util.loss_fn(net, (cg, ng), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
    loss.backward()
    torch.cuda.synchronize()
    ws = [w for k, w in net._engine._ws.items() if k[2]][0]
    snap = {}
    for name, t in ws.t.items():
        if isinstance(t, BNState):
            for f in ("scale", "shift", "mean", "rstd", "ca", "cb", "cc"):
                snap[name + "." + f] = getattr(t, f).double().cpu()
        elif torch.is_tensor(t) and t.is_floating_point() and (name.startswith(("z:", "dy:")) or name in ("hout", "gi", "dgi", "dghn", "dhout", "x")):
            snap[name] = t.clone()
    snaps[mode] = snap
order = sorted(snaps[0], key=lambda n: (n.startswith("dy:") or ".c" in n, n))
for n in order:
    a, b = snaps[1][n], snaps[0][n]
    d = (a - b)
    e = (d.norm() / (b.norm() + 1e-30)).item()
    flag = "  <<<<" if e > 1e-4 else ""
    print("%-16s rel L2 %.2e  max|d| %.2e  max|ref| %.2e%s" % (n, e, d.abs().max().item(), b.abs().max().item(), flag))
