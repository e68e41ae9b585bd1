"""Shared definition of the small 2-rank data-parallel case used by tests/golden/make_dist_golden.py (which runs it
through the REFERENCE's distributed.apply_gradient_allreduce) and tests/test_host_cpu.py (which runs it through ours)."""
import torch
import torch.nn as nn


class Case(nn.Module):
    """a BatchNorm (must stay unsynchronised) and a registered-but-never-called submodule (its gradients stay None and
    must be skipped, like the TGRU block of network.py:150)"""

    def __init__(self):
        super().__init__()
        self.a = nn.Linear(6, 5)
        self.bn = nn.BatchNorm1d(5)
        self.b = nn.Linear(5, 3)
        self.unused = nn.Linear(3, 3)

    def forward(self, x):
        return self.b(self.bn(self.a(x)))


def make(rank):
    torch.manual_seed(100 + rank)          # rank-dependent initial weights: the start-up broadcast must fix them
    return Case()


def inputs(rank):
    return torch.randn(4, 6, generator=torch.Generator().manual_seed(7 + rank))


def run(rank, world, apply_gradient_allreduce, reduce_tensor):
    """one forward/backward through the wrapped module; returns what the fixture stores"""
    net = make(rank)
    apply_gradient_allreduce(net)
    y = net(inputs(rank))
    y.square().sum().backward()
    out = {"state": torch.cat([p.detach().reshape(-1) for p in net.parameters()]).numpy(),
           "grads": torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None]).numpy(),
           "unused_none": all(p.grad is None for p in net.unused.parameters()),
           "bn_mean": net.bn.running_mean.numpy().copy(),
           "reduced": float(reduce_tensor(torch.tensor([float(rank + 1)]), world))}
    return out
