"""Soak: the full-size train step (64 x 4 s) repeated from the SAME weights and batch must give bit-identical gradients and
loss every time (no float atomics, fixed reduction orders); a rare race in a kernel shows up as a differing checksum.
  python scripts/dbg/soak_determinism.py [reps] [dtype]"""
import hashlib
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinyrecurrentunet_amd import network as hn, stft_loss as sl, util

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dev = "cuda"
CFG = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200], sc_lambda=0.5, mag_lambda=0.5,
           band="full")
g = torch.Generator(device=dev); g.manual_seed(11)
B, L = 64, 64000
c = 0.1 * torch.randn((B, 1, L + 1), generator=g, device=dev)
clean = (0.5 * (c[..., 1:] + c[..., :-1])).contiguous()
noisy = (clean + 0.05 * torch.randn((B, 1, L), generator=g, device=dev)).contiguous()
mr = sl.MultiResolutionSTFTLoss(**CFG).to(dev)
torch.manual_seed(0)
net = hn.TRUNet(input_size=4, precision=prec).to(dev).train()
seen = {}
for it in range(reps):
    for p in net.parameters():
        p.grad = None
    loss, _ = util.loss_fn(net, (clean, noisy), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
    loss.backward()
    h = hashlib.sha1()
    h.update(loss.detach().cpu().numpy().tobytes())
    for n, p in net.named_parameters():
        if p.grad is not None:
            h.update(p.grad.detach().cpu().numpy().tobytes())
    seen.setdefault(h.hexdigest(), []).append(it)
print(prec, "reps", reps, "distinct results", len(seen), {k[:10]: (len(v), v[:5]) for k, v in seen.items()}, "loss %.6f" % float(loss))
sys.exit(0 if len(seen) == 1 else 1)
