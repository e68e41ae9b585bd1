"""Which ATen ops (copies, fills) does one fp32 train step launch besides libtrunet_hip?  torch.profiler, CPU-side op table."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from tinyrecurrentunet_amd import network as hn, optim, stft_loss as sl, util
CFG = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200], sc_lambda=0.5, mag_lambda=0.5, band="full")
dev = "cuda"
prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
torch.manual_seed(0)
net = hn.TRUNet(input_size=4, precision=prec).to(dev).train()
opt = optim.FusedAdamW(net.parameters(), lr=4e-4)
sched = util.LinearWarmupCosineDecay(opt, lr_max=4e-4, n_iter=25000000, iteration=0, divider=25, warmup_proportion=0.05, phase=("linear", "cosine"))
mr = sl.MultiResolutionSTFTLoss(**CFG).to(dev)
clean = 0.1 * torch.randn(8, 1, 64000, device=dev); noisy = clean + 0.05 * torch.randn(8, 1, 64000, device=dev)
def step():
    opt.zero_grad()
    loss, _ = util.loss_fn(net, (clean, noisy), ell_p=1, ell_p_lambda=1, stft_lambda=1, mrstftloss=mr)
    loss.backward(); sched.step(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
evs = [e for e in prof.events() if "emcpy" in e.name or "copyBuffer" in e.name]
print("memcpy-like events:", len(evs))
import collections
c = collections.Counter()
for e in prof.events():
    if e.name.startswith("hipMemcpy") or "Memcpy" in e.name:
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 4:
            chain.append(par.name); par = par.cpu_parent
        c[(e.name, " <- ".join(chain))] += 1
for k, v in c.most_common(15):
    print(v, k)
print(prof.key_averages().table(sort_by="count", row_limit=25, max_name_column_width=40))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=40, max_name_column_width=30, max_src_column_width=90))
