"""Layer-by-layer A/B of the training forward at the benchmarked size with the bf16-split path on and off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tinyrecurrentunet_amd import _lib, network as hn
from tinyrecurrentunet_amd.engine import TRUNetEngine
lib = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32064
torch.manual_seed(0)
net = hn.TRUNet(input_size=4).cuda().train()
x = torch.randn(N, 4, 257, device="cuda") * 0.5
res = {}
for mode in (0, 1, 1):
    lib.trunet_gemm_x3_enable(mode)
    eng = TRUNetEngine(net)
    out, (acts, _, NP, w, _) = eng.forward(x, True, record=True)
    torch.cuda.synchronize()
    cur = {k: (a.t[:, :, :N].clone(), None if a.bn is None else (a.bn.scale.clone(), a.bn.shift.clone())) for k, a in acts.items() if hasattr(a, "t")}
    if mode in res:
        # repeatability of the split path
        bad = [k for k in cur if not torch.equal(cur[k][0], res[mode][k][0])]
        print("repeat run identical:", not bad, bad[:5])
    res[mode] = cur
for k in res[0]:
    a, b = res[1][k][0], res[0][k][0]
    d = (a - b).abs()
    line = "%-10s rel L2 %.2e  max %.2e of max|z| %.2e" % (k, (d.norm() / b.norm()).item(), d.max().item(), b.abs().max().item())
    if res[0][k][1] is not None:
        sa, sb = res[1][k][1][0], res[0][k][1][0]
        line += "   BN scale rel %.2e" % ((sa - sb).norm() / sb.norm()).item()
    print(line)
