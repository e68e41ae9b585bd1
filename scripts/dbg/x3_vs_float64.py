"""Which of the two fp32 paths is closer to the truth?  The configs[1]-shaped step (B x 4 s) with the bf16-split kernels on
and off against the FLOAT64 oracle on the host: net output and all parameter gradients.
    python scripts/dbg/x3_vs_float64.py 32 > gpurun_out/x3_vs_float64.txt"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import features_ref as fr, loss_ref, network_ref as nr, weights as W
from tinyrecurrentunet_amd import _lib, dataset as ds, network as hn, stft_loss as sl, util
lib = _lib.lib()
CFG = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200], sc_lambda=0.5, mag_lambda=0.5, band="full")
B, L = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 64000
clean, noisy = W.synth_pairs(B, L, seed=1234)
ref32 = W.fill_state_dict(nr.TRUNet(input_size=4), seed=0)
net = hn.TRUNet(input_size=4)
net.load_state_dict(ref32.state_dict())
net.cuda().train()
mr = sl.MultiResolutionSTFTLoss(**CFG).cuda()
cg, ng = clean.cuda(), noisy.cuda()
res = {}
for mode in (0, 1):
    lib.trunet_gemm_x3_enable(mode)
    net.load_state_dict(ref32.state_dict())
    net.zero_grad()
    loss, info = util.loss_fn(net, (cg, ng), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
    loss.backward()
    with torch.no_grad():
        feats = ds.stft_features(ng[:, 0].contiguous(), pcen=True)
        y = net(feats).double().cpu()
    torch.cuda.synchronize()
    res[mode] = (float(loss), {n: p.grad.double().cpu() for n, p in net.named_parameters() if p.grad is not None}, y)
torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
t0 = time.time()
refd = W.fill_state_dict(nr.TRUNet(input_size=4), seed=0).double().train()
feats64 = ds.stft_features(ng[:, 0].contiguous(), pcen=True).double().cpu()      # the SAME features for the body comparison
with torch.no_grad():
    y64 = refd(feats64)
refd2 = W.fill_state_dict(nr.TRUNet(input_size=4), seed=0).double().train()
loss64, info64, _ = loss_ref.loss_fn(refd2, clean.double(), noisy.double(), stft_config=CFG, pcen=True)
loss64.backward()
g64 = {n: p.grad for n, p in refd2.named_parameters() if p.grad is not None}
print("float64 oracle: %.0f s; loss %.8f, fp32-MFMA %.8f, split %.8f" % (time.time() - t0, float(loss64), res[0][0], res[1][0]))
for mode, nm in ((0, "fp32-MFMA"), (1, "split    ")):
    y = res[mode][2]
    print("net output vs float64 (same features): %s relative L2 %.2e  max %.2e of max|y|" % (
        nm, ((y - y64).norm() / y64.norm()).item(), ((y - y64).abs().max() / y64.abs().max()).item()))
def dist(a, b):
    es = []
    for n, x in a.items():
        y = b[n]
        if y.abs().max().item() < 1e-3:
            continue
        es.append(((x - y).norm() / y.norm()).item())
    return "median %.2e max %.2e (%d tensors)" % (np.median(es), max(es), len(es))
print("B = %d gradients, relative L2 per tensor:" % B)
print("  fp32-MFMA vs float64 : " + dist(res[0][1], g64))
print("  split     vs float64 : " + dist(res[1][1], g64))
print("  split     vs fp32-MFMA: " + dist(res[1][1], res[0][1]))
lib.trunet_gemm_x3_enable(1)
