"""Which fp32 implementation is how far from the fp64 oracle, per parameter gradient?
Columns: HIP (ours), torch CPU fp32 (double accumulators inside BN), torch GPU fp32 (MIOpen/ATen)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import network_ref as nr, weights as W
from tinyrecurrentunet_amd import network as hn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
x = torch.tensor(np.random.default_rng(N).standard_normal((N, 4, 257)) * 0.5, dtype=torch.float32)
cot = torch.tensor(np.random.default_rng(N + 1).standard_normal((N, 8, 257)), dtype=torch.float32)
mk = lambda: W.fill_state_dict(nr.TRUNet(input_size=4), seed=seed)
refd = mk().double().train(); y = refd(x.double()); (y * cot.double()).sum().backward()
r32 = mk().train(); y = r32(x.clone()); (y * cot).sum().backward()
g32 = mk().cuda().train(); y = g32(x.cuda()); (y * cot.cuda()).sum().backward()
net = hn.TRUNet(input_size=4); net.load_state_dict(mk().state_dict()); net.cuda().train()
y = net(x.cuda()); (y * cot.cuda()).sum().backward()
pd, p32, pg = dict(refd.named_parameters()), dict(r32.named_parameters()), dict(g32.named_parameters())
def e(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
print("%-52s %10s %10s %10s" % ("param (rel L2 err vs f64)", "HIP", "torchCPU32", "torchGPU32"))
rows = []
for n, p in net.named_parameters():
    if n.startswith("TGRU") or float(pd[n].grad.abs().max()) < 1e-6:
        continue
    rows.append((n, e(p.grad, pd[n].grad), e(p32[n].grad, pd[n].grad), e(pg[n].grad, pd[n].grad)))
    print("%-52s %10.2e %10.2e %10.2e" % rows[-1])
a = np.array([[r[1], r[2], r[3]] for r in rows])
print("median", np.median(a, 0), "max", a.max(0))
