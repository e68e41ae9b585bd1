"""Loss curves of a few optimisation steps on one fixed synthetic batch: fp32 vs bf16 storage (same initial weights)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinyrecurrentunet_amd import network as hn, optim, stft_loss as sl, util
CFG = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200], sc_lambda=0.5, mag_lambda=0.5,
           band="full")
dev = "cuda"
SEED = int(os.environ.get("CURVE_SEED", "0"))
g = torch.Generator(device=dev); g.manual_seed(3 + SEED)
B, L = 8, 32000
c = 0.1 * torch.randn((B, 1, L + 1), generator=g, device=dev)
clean = (0.5 * (c[..., 1:] + c[..., :-1])).contiguous()
noisy = (clean + 0.05 * torch.randn((B, 1, L), generator=g, device=dev)).contiguous()
mr = sl.MultiResolutionSTFTLoss(**CFG).to(dev)
curves = {}
for prec in ("fp32", "bf16"):
    torch.manual_seed(SEED)
    net = hn.TRUNet(input_size=4, precision=prec).to(dev).train()
    opt = optim.FusedAdamW(net.parameters(), lr=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3)
    ls = []
    for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
        opt.zero_grad()
        loss, info = util.loss_fn(net, (clean, noisy), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
        loss.backward()
        opt.step()
        ls.append(float(loss))
    curves[prec] = ls
step = int(os.environ.get("CURVE_EVERY", "5"))
for i in range(0, len(curves["fp32"]), step):
    print("%3d  fp32 %.4f   bf16 %.4f   %+.4f" % (i, curves["fp32"][i], curves["bf16"][i], curves["bf16"][i] / curves["fp32"][i] - 1))
print("last fp32 %.4f bf16 %.4f" % (curves["fp32"][-1], curves["bf16"][-1]))
