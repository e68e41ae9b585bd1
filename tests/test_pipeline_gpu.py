"""GPU: the rows either side of the hot path -- input pipeline, fused optimizer tail with torch-format
checkpoints (8f rank 3, a16), and the exchange step on RCCL (a15)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(fft_sizes=[512, 1024, 2048], hop_sizes=[50, 120, 240], win_lengths=[240, 600, 1200],
           sc_lambda=0.5, mag_lambda=0.5, band="full")


@pytest.mark.parametrize("B,L", [(1, 1000), (5, 32000), (3, 96000), (2, 257)])
def test_augment_mix_vs_oracle(B, L):
    """trunet_augment_mix (gain -> low-pass biquad -> clamp -> high-pass biquad -> clamp, + clean) vs the float64
    scipy.lfilter restatement of torchaudio's published biquads (oracle/augment_ref.py; torchaudio itself is absent:
    parity unpinned by the reference for this stage).  The chunked recurrence must agree with the sequential one."""
    from oracle import augment_ref as ar
    from tinyrecurrentunet_amd import dataset as ds
    rng = np.random.default_rng(B * L)
    noise = (rng.standard_normal((B, 1, L)) * 0.3).astype(np.float32)
    noise[0, 0, : min(50, L)] = 3.0                               # drive the clamp
    clean = (rng.standard_normal((B, 1, L)) * 0.1).astype(np.float32)
    aug = ds.DataAugment()
    draws = [(7000 + 100 * (3 * b % 30), 800 + 50 * (b % 8), -12 + 0.033 * (17 * b % 200)) for b in range(B)]
    params = np.stack([aug.params(*d) for d in draws])
    ref = np.stack([ar.augment(noise[b, 0], 48000, d[2], d[0], d[1]) for b, d in enumerate(draws)])
    got = aug(torch.tensor(noise).cuda(), params=params).cpu().numpy()[:, 0]
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()
    assert np.abs(ref).max() > 0.05
    # through the loader's staging path: noisy = clean + augmented noise
    from tinyrecurrentunet_amd import _lib as Lb
    nz, cl, par = torch.tensor(noise).cuda(), torch.tensor(clean).cuda(), torch.tensor(params).cuda()
    out = torch.empty_like(nz)
    Lb.check(Lb.lib().trunet_augment_mix(Lb.ptr(nz), Lb.ptr(cl), Lb.ptr(par), Lb.ptr(out), None, B, L, Lb.stream()))
    assert np.abs(out.cpu().numpy()[:, 0] - (clean[:, 0] + ref)).max() < 2e-5


def test_loader_yields_reference_shaped_batches_on_the_gpu(tmp_path):
    """load_CleanNoisyPairDataset end to end on wav files (dataset.py:393-412 + train.py:121-125): DataLoader workers
    read + crop, the batch is staged through pinned memory on a side stream and mixed on the GPU; every batch equals
    clean + oracle-augmented noise for the parameters drawn."""
    from scipy.io.wavfile import write as wavwrite
    from oracle import augment_ref as ar
    from tinyrecurrentunet_amd import dataset as ds
    sr, crop = 16000, 0.5
    (tmp_path / "clean").mkdir()
    (tmp_path / "keyboard").mkdir()
    rng = np.random.default_rng(1)
    for i in range(6):
        wavwrite(str(tmp_path / "clean" / ("fileid_%d.wav" % i)), sr, (rng.standard_normal(sr) * 3000).astype(np.int16))
    noise = (rng.standard_normal(int(sr * crop)) * 6000).astype(np.int16)
    wavwrite(str(tmp_path / "keyboard" / "k0.wav"), sr, noise)
    loader = ds.load_CleanNoisyPairDataset(root=str(tmp_path), subset="training", crop_length_sec=crop, batch_size=4,
                                           sample_rate=sr, num_gpus=1, num_workers=2)
    seen = 0
    for clean, noisy, fileid in loader:
        assert clean.is_cuda and noisy.is_cuda and clean.shape == noisy.shape and clean.shape[1:] == (1, 8000)
        assert len(fileid) == clean.shape[0]
        clean2 = clean.cuda()                                   # train.py:124: a no-op now
        assert clean2.data_ptr() == clean.data_ptr()
        res = (noisy - clean).cpu().numpy()[:, 0]               # the augmented noise: bounded by the clamp, not silent
        assert np.abs(res).max() <= 1.0 + 1e-6 and np.abs(res).max() > 1e-3
        seen += clean.shape[0]
    assert seen == 6
    # exact check of one batch with known parameters (no shuffling, no workers)
    d = loader.dataset
    import random
    random.seed(3)
    np.random.seed(3)
    items = [d[i] for i in range(3)]
    batch = ds._collate_pairs(items)
    side = torch.cuda.Stream()
    clean_d, noisy_d, _, ev = loader._stage(batch, side)
    torch.cuda.current_stream().wait_event(ev)
    nz = noise.astype(np.float32) / 32768.0
    for b, it in enumerate(items):
        p = it[3].numpy().astype(np.float64)
        x = nz.astype(np.float64) * p[0]
        from scipy.signal import lfilter
        x = np.clip(lfilter(p[1:4], np.r_[1.0, p[4:6]], x), -1, 1)
        x = np.clip(lfilter(p[6:9], np.r_[1.0, p[9:11]], x), -1, 1)
        assert np.abs(noisy_d[b, 0].cpu().numpy() - (it[0][0].numpy() + x)).max() < 2e-5


def _train_pair(cin=4, seed=0):
    from oracle import network_ref as nr, weights as W
    from tinyrecurrentunet_amd import network as hn
    ref = W.fill_state_dict(nr.TRUNet(input_size=cin), seed=seed)
    net = hn.TRUNet(input_size=cin)
    net.load_state_dict(ref.state_dict())
    return ref, net.cuda().train()


def test_fused_adamw_on_engine_layout_matches_torch_and_checkpoints():
    """Three train steps of the HIP TRU-Net with FusedAdamW on the engine's flat gradient buffer (no packing) vs
    torch.optim.AdamW fed the very same gradients; then the optimizer_state_dict round trip of train.py:155-162 both
    ways, and a resumed run continues exactly like an uninterrupted one."""
    from tinyrecurrentunet_amd import optim
    _, net = _train_pair()
    params = [p for p in net.parameters()]
    opt = optim.FusedAdamW(net.parameters(), lr=4e-4)
    shadow = [p.detach().clone().requires_grad_(True) for p in params]
    topt = torch.optim.AdamW(shadow, lr=4e-4)
    rng = np.random.default_rng(0)
    cot = torch.tensor(rng.standard_normal((40, 8, 257)), dtype=torch.float32).cuda()
    for it in range(3):
        x = torch.tensor(rng.standard_normal((40, 4, 257)) * 0.5, dtype=torch.float32).cuda()
        opt.zero_grad()
        (net(x) * cot).sum().backward()
        for p, q in zip(params, shadow):
            q.grad = None if p.grad is None else p.grad.detach().clone()
        nsq = opt.step()
        topt.step()
        tot = sum(float((q.grad.double() ** 2).sum()) for q in shadow if q.grad is not None)
        assert abs(float(nsq) - tot) < 1e-4 * tot
        assert opt._key[0] == "engine"                       # the gradients were used where the backward left them
    for (n, p), q in zip(net.named_parameters(), shadow):
        assert float((p.detach() - q.detach()).abs().max()) <= 1e-6 * (1 + float(q.detach().abs().max())), n
    # checkpoint: ours -> torch, torch -> ours
    sd = opt.state_dict()
    t2 = torch.optim.AdamW(shadow, lr=4e-4)
    t2.load_state_dict(sd)
    idx = {i: p for i, p in enumerate(shadow)}
    for i, st in topt.state_dict()["state"].items():
        assert float(st["step"]) == float(sd["state"][i]["step"]) == 3.0
        for k in ("exp_avg", "exp_avg_sq"):       # same recurrences, fp32 rounding of a differently associated product
            assert float((st[k] - sd["state"][i][k]).abs().max()) <= 1e-4 * float(st[k].abs().max()) + 1e-30, (i, k)
    assert sorted(sd["state"]) == sorted(topt.state_dict()["state"])           # TGRU parameters: no state in either
    # resume: a fresh optimizer loaded from the checkpoint takes the same 4th step as the original
    import copy
    net_b = copy.deepcopy(net)
    opt_b = optim.FusedAdamW(net_b.parameters(), lr=1e-3)
    opt_b.load_state_dict(copy.deepcopy(sd))
    x = torch.tensor(rng.standard_normal((40, 4, 257)) * 0.5, dtype=torch.float32).cuda()
    for n_, o_ in ((net, opt), (net_b, opt_b)):
        o_.zero_grad()
        (n_(x) * cot).sum().backward()
        o_.step()
    for (n, p), (_, q) in zip(net.named_parameters(), net_b.named_parameters()):
        assert torch.equal(p.detach(), q.detach()), n


def test_fused_adamw_packed_layout_and_changing_active_set():
    """Gradients that are NOT the engine's views take the packing path; when the set of parameters with gradients
    changes, moments and per-parameter step counts carry over -- compared with torch.optim.AdamW."""
    from tinyrecurrentunet_amd import optim
    torch.manual_seed(0)
    ps = [torch.randn(7, 5, device="cuda", requires_grad=True), torch.randn(11, device="cuda", requires_grad=True),
          torch.randn(3, 3, device="cuda", requires_grad=True)]
    qs = [p.detach().clone().requires_grad_(True) for p in ps]
    a = optim.FusedAdamW(ps, lr=4e-4)
    b = torch.optim.AdamW(qs, lr=4e-4)
    sets = [(0, 1), (0, 1), (0, 1, 2), (1, 2), (0, 1, 2)]
    for it, act in enumerate(sets):
        for i, (p, q) in enumerate(zip(ps, qs)):
            if i in act:
                g = torch.randn_like(p)
                p.grad, q.grad = g.clone(), g.clone()
            else:
                p.grad, q.grad = None, None
        a.step()
        b.step()
        assert a._key[0] == "packed"
    for p, q in zip(ps, qs):
        assert float((p - q).abs().max()) < 1e-6
    sa, sb = a.state_dict()["state"], b.state_dict()["state"]
    for i in range(3):
        assert float(sa[i]["step"]) == float(sb[i]["step"])
        assert float((sa[i]["exp_avg"] - sb[i]["exp_avg"]).abs().max()) < 1e-7


def test_rccl_world_size_one_allreduce_on_the_flat_gradient():
    """init_distributed (distributed.py:48-58) with backend "nccl" (= RCCL) and world size 1, apply_gradient_allreduce on
    the HIP TRU-Net, one backward: the collective runs on the engine's flat gradient tensor in place.  (One GPU on this
    box: 2-rank semantics are covered over gloo in test_dp_gpu.py / test_host_cpu.py, the 1 -> 8 curve is the driver's.)"""
    code = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from oracle import network_ref as nr, weights as W
from tinyrecurrentunet_amd import distributed as td, network as hn
td.init_distributed(0, 1, "g", "nccl", "tcp://127.0.0.1:%%d" %% int(sys.argv[1]))
assert dist.get_backend() == "nccl"
sd = W.fill_state_dict(nr.TRUNet(input_size=4), seed=3).state_dict()
net = hn.TRUNet(input_size=4); net.load_state_dict(sd); net.cuda().train()
ref = hn.TRUNet(input_size=4); ref.load_state_dict(sd); ref.cuda().train()
td.apply_gradient_allreduce(net)
x = torch.tensor(np.random.default_rng(0).standard_normal((40, 4, 257)) * 0.5, dtype=torch.float32).cuda()
cot = torch.tensor(np.random.default_rng(1).standard_normal((40, 8, 257)), dtype=torch.float32).cuda()
(net(x) * cot).sum().backward()
(ref(x) * cot).sum().backward()
torch.cuda.synchronize()
assert net._grad_bucket.in_place is True, net._grad_bucket.in_place
n = 0
for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
    if q.grad is None:
        assert p.grad is None, k
        continue
    assert torch.equal(p.grad, q.grad), k          # mean over one rank = the local gradient, bit for bit
    n += p.numel()
assert n == 298592
r = td.reduce_tensor(torch.tensor([2.5], device="cuda"), 1)
assert float(r) == 2.5
dist.destroy_process_group()
print("rccl ok")
''' % ROOT
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", code, str(port)], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and "rccl ok" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])
