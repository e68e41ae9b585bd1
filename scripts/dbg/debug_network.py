"""Layer-by-layer comparison of the HIP schedule with the oracle (diagnostic; run on the GPU box):
    python scripts/dbg/debug_network.py [N] > gpurun_out/debug_network.txt
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import network_ref as nr, weights as W  # noqa: E402
from tinyrecurrentunet_amd import network as hn  # noqa: E402


def rel(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 37
    cin = 4
    torch.manual_seed(0)
    ref = W.fill_state_dict(nr.TRUNet(input_size=cin), seed=0)
    net = hn.TRUNet(input_size=cin)
    net.load_state_dict(ref.state_dict())
    net.cuda()
    x = torch.tensor(np.random.default_rng(1).standard_normal((N, cin, 257)) * 0.7, dtype=torch.float32)
    for mode in ("eval", "train"):
        ref.train(mode == "train")
        net.train(mode == "train")
        refd = W.fill_state_dict(nr.TRUNet(input_size=cin), seed=0).double().train(mode == "train")
        with torch.no_grad():
            yr, inter = ref(x.clone(), return_intermediates=True)
            yd, interd = refd(x.double(), return_intermediates=True)
        xg = x.cuda()
        if mode == "train":
            xg.requires_grad_(False)
        y = net(xg)
        acts = net._engine._ws[list(net._engine._ws)[0]].t
        eng_acts = None
        print("== mode", mode, "N", N)
        # reconstruct post-activation tensors from the engine's raw buffers
        ws = net._engine._ws[list(net._engine._ws)[0]]
        for name in ["enc0", "enc1", "enc2", "enc3", "enc4", "enc5", "fgru", "dec0", "dec1", "dec2", "dec3", "dec4",
                     "dec5"]:
            z = ws.t["z:" + name][:, :, :N].permute(2, 0, 1).cpu()
            bn = ws.t.get("bn:" + name)
            if name == "enc0" or name == "dec5":
                a = z
            else:
                a = torch.relu(z * bn.scale.cpu()[None, :, None] + bn.shift.cpu()[None, :, None])
            print("%-6s shape %-18s rel_err_vs_f64 %.3e   (f32 torch vs f64: %.3e)" % (
                name, tuple(a.shape), rel(a, interd[name]), rel(inter[name], interd[name])))
        print("out    rel_err_vs_f64 %.3e   (f32 torch vs f64: %.3e)" % (rel(y.cpu(), yd), rel(yr, yd)))
    # gradients
    ref.train(); net.train()
    refd = W.fill_state_dict(nr.TRUNet(input_size=cin), seed=0).double().train()
    cot = torch.tensor(np.random.default_rng(2).standard_normal((N, 8, 257)), dtype=torch.float32)
    yd = refd(x.double()); (yd * cot.double()).sum().backward()
    yr = ref(x.clone()); (yr * cot).sum().backward()
    y = net(x.cuda()); (y * cot.cuda()).sum().backward()
    print("== gradients (rel to max |g| of the f64 oracle)")
    pr = dict(ref.named_parameters()); pd = dict(refd.named_parameters())
    worst = 0.0
    for n, p in net.named_parameters():
        if n.startswith("TGRU"):
            continue
        if p.grad is None:
            print("%-55s MISSING" % n); continue
        e = rel(p.grad.cpu(), pd[n].grad); e32 = rel(pr[n].grad, pd[n].grad)
        worst = max(worst, e)
        print("%-55s %.3e  (torch f32: %.3e)  |g|max %.3e" % (n, e, e32, float(pd[n].grad.abs().max())))
    print("worst", worst)
    for n, b in net.named_buffers():
        if "running" in n and not n.startswith("TGRU"):
            e = rel(b.cpu(), dict(refd.named_buffers())[n])
            if e > 1e-5:
                print("buffer", n, e)


if __name__ == "__main__":
    main()
