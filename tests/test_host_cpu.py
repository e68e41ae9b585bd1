"""CPU: host-side logic, the C-ABI surface, and the 2-rank data-parallel path on gloo."""
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_abi_exports_every_declared_symbol():
    """libtrunet_hip.so loads without a GPU and exports exactly what include/trunet_hip.h declares."""
    import ctypes
    from tinyrecurrentunet_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "trunet_hip.h")).read()
    declared = set(re.findall(r"^\s*int\s+(trunet_\w+)\s*\(", hdr, flags=re.M))
    assert len(declared) >= 29
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert set(_lib.declared_symbols()) == declared
    # struct layouts agree with the header (sizes as hipcc lays them out)
    assert ctypes.sizeof(_lib.Seg) == 5 * 8 + 8 * 4
    assert ctypes.sizeof(_lib.GemmArgs) == 14 * 4 + 8 * 8 + 5 * ctypes.sizeof(_lib.Seg)
    assert ctypes.sizeof(_lib.WgradArgs) == 16 * 4 + 7 * 8 + 2 * 4 + 5 * ctypes.sizeof(_lib.Seg)
    assert ctypes.sizeof(_lib.DgradOut) == 4 * 8 + 2 * 4
    assert ctypes.sizeof(_lib.PwBwdArgs) == ctypes.sizeof(_lib.WgradArgs) + 8 + 5 * ctypes.sizeof(_lib.DgradOut)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "tinyrecurrentunet_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("oracle/", ""), fn


def test_scheduler_matches_reference(golden):
    from tinyrecurrentunet_amd import util

    class Opt:
        param_groups = [{"lr": 0.0}]
    s = util.LinearWarmupCosineDecay(Opt(), lr_max=4e-4, n_iter=1000, iteration=0, divider=25,
                                     warmup_proportion=0.05, phase=("linear", "cosine"))
    lrs = np.array([s.step() for _ in range(1000)])
    np.testing.assert_allclose(lrs, golden("sched")["lrs"], rtol=1e-12, atol=0)
    assert Opt.param_groups[0]["lr"] == lrs[-1]
    # resume mid-way (train.py:102-110 passes iteration=n_iter)
    s2 = util.LinearWarmupCosineDecay(Opt(), lr_max=4e-4, n_iter=1000, iteration=300, divider=25,
                                      warmup_proportion=0.05)
    assert abs(s2.step() - lrs[300]) < 1e-15


def test_find_max_epoch_and_misc(tmp_path):
    from tinyrecurrentunet_amd import util
    assert util.find_max_epoch(str(tmp_path)) == -1
    for n in ("10.pkl", "250.pkl", "x.pkl", "7.txt"):
        (tmp_path / n).write_text("")
    assert util.find_max_epoch(str(tmp_path)) == 250
    r = util.rescale(torch.tensor([1.0, 3.0, 2.0]))
    assert float(r.min()) == 0 and float(r.max()) == 1


def test_engine_launch_mirror():
    from tinyrecurrentunet_amd import _lib, engine
    mk = lambda n: type("S", (), {"nchan": n})()
    assert engine._gemm_rs(128, [mk(128)]) == 4
    assert engine._gemm_rs(64, [mk(64), mk(128)]) == 2
    assert engine._gemm_rs(8, [mk(64), mk(64)]) == 1
    assert engine._gemm_rs(128, [mk(384)]) == 2        # weight block capped at 112 KiB of LDS
    assert engine.ceil_to(32064, engine.FRAME_PAD) == 32256 and engine.ceil_to(256, 256) == 256


def test_state_dict_round_trip_with_oracle_layout():
    from oracle import network_ref as nr, weights as W
    from tinyrecurrentunet_amd import network as hn
    ref = W.fill_state_dict(nr.TRUNet(input_size=4), seed=1)
    net = hn.TRUNet(input_size=4)
    net.load_state_dict(ref.state_dict())
    back = nr.TRUNet(input_size=4)
    back.load_state_dict(net.state_dict())
    assert abs(W.checksum(back) - W.checksum(ref)) < 1e-9
    assert sum(p.numel() for n, p in net.named_parameters() if not n.startswith("TGRU")) == 298592


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from tinyrecurrentunet_amd import distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank: broadcast must fix it
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3),
                              torch.nn.Linear(3, 3))   # last layer unused -> grad None (like TGRU, R4)
    td.apply_gradient_allreduce(net)
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(7 + rank))

    def fwd(m, inp):
        return m[2](m[1](m[0](inp)))
    net.register_forward_hook(lambda *a: None)
    y = net[2](net[1](net[0](x)))
    net.needs_reduction = True                          # armed by the module forward hook in real use
    y.square().sum().backward()
    state = torch.cat([p.detach().reshape(-1) for p in list(net.parameters())[:4]])
    grads = torch.cat([p.grad.reshape(-1) for p in net.parameters() if p.grad is not None])
    rt = td.reduce_tensor(torch.tensor([float(rank + 1)]), world)
    q.put((rank, state.numpy(), grads.numpy(), net[3].weight.grad is None, float(rt), x.numpy()))
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_gloo():
    """distributed.py:95-147 semantics: state broadcast from rank 0, gradients averaged once per backward,
    parameters without gradient skipped, BatchNorm not synchronised."""
    world, port = 2, 29000 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    [p.join(60) for p in ps]
    (r0, s0, g0, none0, rt0, x0), (r1, s1, g1, none1, rt1, x1) = res
    np.testing.assert_array_equal(s0, s1)               # identical parameters after the broadcast
    np.testing.assert_array_equal(g0, g1)               # identical (averaged) gradients
    assert none0 and none1
    assert rt0 == rt1 == 1.5
    # reference value: average of the two single-rank gradients computed from rank 0's parameters
    torch.manual_seed(100)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.BatchNorm1d(5), torch.nn.Linear(5, 3), torch.nn.Linear(3, 3))
    gs = []
    for x in (x0, x1):
        net.zero_grad()
        import copy
        m = copy.deepcopy(net)
        y = m[2](m[1](m[0](torch.tensor(x))))
        y.square().sum().backward()
        gs.append(torch.cat([p.grad.reshape(-1) for p in m.parameters() if p.grad is not None]).numpy())
    np.testing.assert_allclose(g0, 0.5 * (gs[0] + gs[1]), rtol=1e-5, atol=1e-6)


def test_checkpoint_round_trip_with_reference_layout(tmp_path):
    """train.py:155-162 checkpoints ({'iter','model_state_dict','optimizer_state_dict','training_time_seconds'} as
    '<iter>.pkl') load both ways between the reference layout (oracle modules = the reference's own block classes in the
    R1 composition) and the HIP-backed modules: same 177 keys, same shapes, strict load; util.find_max_epoch finds it."""
    from oracle import network_ref as nr, weights as W
    from tinyrecurrentunet_amd import network as hn, util
    ref = W.fill_state_dict(nr.TRUNet(input_size=4), seed=21)
    opt = torch.optim.AdamW(ref.parameters(), lr=4e-4)
    path = tmp_path / "1200.pkl"
    torch.save({"iter": 1200, "model_state_dict": ref.state_dict(), "optimizer_state_dict": opt.state_dict(),
                "training_time_seconds": 12}, str(path))
    assert util.find_max_epoch(str(tmp_path)) == 1200
    ck = torch.load(str(path), map_location="cpu", weights_only=True)
    ours = hn.TRUNet(input_size=4)                       # parameter containers only: no GPU needed to load / save
    missing = ours.load_state_dict(ck["model_state_dict"], strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    sd = ours.state_dict()
    assert len(sd) == 177 and list(sd) == list(ref.state_dict())
    for k, v in ref.state_dict().items():
        assert torch.equal(sd[k], v), k
    # and back: a checkpoint written from our module loads into the reference layout
    torch.save({"iter": 1, "model_state_dict": ours.state_dict()}, str(tmp_path / "1.pkl"))
    back = nr.TRUNet(input_size=4)
    back.load_state_dict(torch.load(str(tmp_path / "1.pkl"), map_location="cpu", weights_only=True)["model_state_dict"])
    assert W.checksum(back) == W.checksum(ref)
    # the use_tgru extension does not change the parameter set
    assert list(hn.TRUNet(input_size=4, use_tgru=True).state_dict()) == list(sd)
