"""How sensitive is the whole-network gradient to bf16-sized perturbations?  fp32 HIP path with bf16-ROUNDED WEIGHTS (one of
the three roundings the bf16 path applies per layer) against the fp32 HIP path with exact weights, next to bf16 vs fp32."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tinyrecurrentunet_amd.network import TRUNet

def l2(a, b):
    return ((a - b).norm() / (b.norm() + 1e-30)).item()

torch.manual_seed(0)
N, cin = 501, 4
f32 = TRUNet(input_size=cin).cuda().train()
f32w = TRUNet(input_size=cin).cuda().train()
b16 = TRUNet(input_size=cin, precision="bf16").cuda().train()
f32w.load_state_dict(f32.state_dict()); b16.load_state_dict(f32.state_dict())
with torch.no_grad():
    for p in f32w.parameters():
        if p.dim() > 1:
            p.copy_(p.bfloat16().float())
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = torch.randn(N, cin, 257, generator=g, device="cuda")
mode = sys.argv[1] if len(sys.argv) > 1 else "rand"
if mode == "rand":
    gout = torch.randn(N, 8, 257, generator=g, device="cuda") / N
outs = []
for net in (f32, f32w, b16):
    y = net(x)
    if mode == "rand":
        y.backward(gout)
    else:
        (y.square().mean()).backward()      # a smooth loss
    outs.append(y.detach())
print("forward  rounded-weights vs fp32 %.3e   bf16 vs fp32 %.3e" % (l2(outs[1], outs[0]), l2(outs[2], outs[0])))
for (n, p), (_, q), (_, r) in zip(f32.named_parameters(), f32w.named_parameters(), b16.named_parameters()):
    if n.startswith("TGRU") or p.grad is None or p.grad.norm() < 1e-6:
        continue
    print("%-48s w-rounded %.3e   bf16 %.3e   bf16 vs w-rounded %.3e" % (n, l2(q.grad, p.grad), l2(r.grad, p.grad), l2(r.grad, q.grad)))
