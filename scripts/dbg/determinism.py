"""Diagnostic: run the same TRU-Net forward/backward repeatedly and compare every gradient bit for bit with the first
run (all kernels are meant to be deterministic: two-stage reductions, no float atomics in the body)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import network_ref as nr, weights as W
from tinyrecurrentunet_amd import network as hn

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
net = hn.TRUNet(input_size=4)
net.load_state_dict(W.fill_state_dict(nr.TRUNet(input_size=4), seed=3).state_dict())
net.cuda().train()
x = torch.tensor(np.random.default_rng(N).standard_normal((N, 4, 257)) * 0.5, dtype=torch.float32).cuda()
cot = torch.tensor(np.random.default_rng(N + 1).standard_normal((N, 8, 257)), dtype=torch.float32).cuda()
ref = None
bad = {}
for it in range(reps):
    for p in net.parameters():
        p.grad = None
    y = net(x)
    (y * cot).sum().backward()
    torch.cuda.synchronize()
    cur = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
    cur["__y"] = y.detach().clone()
    ws = list(net._engine._ws.values())[0]
    if os.environ.get("DET_CHECKSUMS"):
        for k, t in ws.t.items():
            if isinstance(t, torch.Tensor) and k.startswith("dy:"):
                cur["ws:" + k] = t[..., :N].double().sum().reshape(1)
            elif k.startswith("bn:"):
                cur["ws:" + k + ".ca"] = t.ca.double().sum().reshape(1)
                cur["ws:" + k + ".cc"] = t.cc.double().sum().reshape(1)
    if ref is None:
        ref = cur
        continue
    for n in cur:
        if not torch.equal(cur[n], ref[n]):
            d = float((cur[n] - ref[n]).abs().max())
            bad.setdefault(n, []).append((it, d, float(ref[n].abs().max())))
if not bad:
    print("deterministic over %d runs (N=%d)" % (reps, N))
else:
    for n, v in bad.items():
        print("NONDETERMINISTIC %-55s runs %s maxdiff %.3g (max|g| %.3g)" % (n, [i for i, _, _ in v], max(d for _, d, _ in v), v[0][2]))
