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
    declared = set(re.findall(r"^\s*(?:int|size_t)\s+(trunet_\w+)\s*\(", hdr, flags=re.M))
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


def test_scheduler_building_blocks_of_the_reference_api(golden):
    """the reference's util module exports anneal_linear / anneal_cosine / Phase (util.py:81-107) and flatten
    (:22-23); a schedule assembled from them the way util.py:110-156 does gives the learning rates of the reference fixture,
    i.e. the same as the closed-form class."""
    from tinyrecurrentunet_amd import util
    assert util.flatten([[1, 2], [3], []]) == [1, 2, 3]
    assert util.anneal_linear(1.0, 3.0, 0.25) == 1.5
    assert abs(util.anneal_cosine(4.0, 0.0, 0.5) - 2.0) < 1e-12 and util.anneal_cosine(4.0, 1.0, 0.0) == 4.0
    lr_max, n_iter, div, warm = 4e-4, 1000, 25, 50
    legs = [util.Phase(lr_max / div, lr_max, warm, 0, util.anneal_linear),
            util.Phase(lr_max, lr_max / div / 1e4, n_iter - warm, 0, util.anneal_cosine)]
    lrs, k = [], 0
    for _ in range(n_iter):
        lrs.append(legs[k].step())
        if legs[k].is_done:
            legs[k].reset()
            k = 1 - k
    np.testing.assert_allclose(np.array(lrs), golden("sched")["lrs"], rtol=1e-12, atol=0)

    class Layer:
        weight = torch.nn.Parameter(torch.arange(12.0).reshape(3, 4))
        bias = torch.nn.Parameter(torch.ones(3))
    w0 = Layer.weight.detach().clone()
    util.weight_scaling_init(Layer)
    a = torch.sqrt(10.0 * w0.std())
    assert torch.allclose(Layer.weight.detach(), w0 / a) and torch.allclose(Layer.bias.detach(), torch.ones(3) / a)


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


def _dist_case_worker(rank, world, port, q):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dist_case
    from tinyrecurrentunet_amd import distributed as td
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q.put((rank, dist_case.run(rank, world, td.apply_gradient_allreduce, td.reduce_tensor)))
    dist.destroy_process_group()


def test_gradient_allreduce_matches_reference_fixture(golden):
    """a15 pinned by the reference itself: tests/golden/dist_allreduce.npz was produced by running tests/dist_case.py
    through /root/reference/distributed.py's own apply_gradient_allreduce / reduce_tensor on 2 gloo ranks
    (make_dist_golden.py); the same case through this build's wrapper must give the same parameters after the
    start-up broadcast, the same averaged gradients, the same per-rank BatchNorm statistics."""
    import socket
    g = golden("dist_allreduce")
    world = 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_dist_case_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    for r in range(world):
        np.testing.assert_array_equal(res[r]["state"], g["r%d_state" % r])
        np.testing.assert_allclose(res[r]["grads"], g["r%d_grads" % r], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(res[r]["bn_mean"], g["r%d_bn_mean" % r], rtol=1e-6)
        assert res[r]["unused_none"] and bool(g["r%d_unused_none" % r])
        assert res[r]["reduced"] == float(g["r%d_reduced" % r]) == 1.5
    assert not np.array_equal(res[0]["bn_mean"], res[1]["bn_mean"])       # BatchNorm stays per rank


def test_dropin_modules_satisfy_the_reference_import_lines():
    """Every ``from <module> import <names>`` the reference's harness scripts execute against the hot-path modules
    (train.py:16-22, rt.py:8, stream.py:14, onnx.py's TRUNet use) resolves against dropin/ -- in a fresh interpreter with
    dropin/ first on sys.path, exactly as INTEGRATION.md tells a maintainer to launch."""
    import subprocess
    lines = ["from distributed import init_distributed, apply_gradient_allreduce, reduce_tensor",     # train.py:16
             "from dataset import load_CleanNoisyPairDataset",                                        # train.py:17
             "from stft_loss import MultiResolutionSTFTLoss",                                         # train.py:18
             "from util import rescale, find_max_epoch, print_size",                                  # train.py:19
             "from util import LinearWarmupCosineDecay, loss_fn",                                     # train.py:20
             "from util import anneal_linear, anneal_cosine, Phase, flatten, std_normal, weight_scaling_init, sampling",
             "from network import TRUNet, TRUNet2D",                                                  # train.py:22
             "from network import TRUNet",                                                            # rt.py:8
             "from dataset import ProcessAudio",                                                      # stream.py:14
             "from dataset import CleanNoisyPairDataset, DataAugment, pcenfunc, unwrap",
             "from phm import PhaseAwareMask",                                                        # network.py:6
             "from network import StandardConv1d, DepthwiseSeparableConv1d, GRUBlock, FirstTrCNN, TrCNN, LastTrCNN",
             "from stft_loss import stft, SpectralConvergenceLoss, LogSTFTMagnitudeLoss, STFTLoss",
             "dp = ProcessAudio(); assert all(hasattr(dp, n) for n in ('mod_phase', 'get_mag_phase', 'demod_phase', "
             "'perm', 'de_perm', 'norm', 'de_norm', 'amp_to_db', 'db_to_amp', 'forward', 'backward'))",
             # config/tiny.json's sections unpacked as **kwargs exactly like train.py:54,61,114,131 / :180-192
             "import json; cfg = json.load(open(%r)); net = TRUNet(**cfg['network']); "
             "assert len(net.state_dict()) == 177" % os.path.join(ROOT, "tests", "golden", "tiny_config_sections.json"),
             "mr = MultiResolutionSTFTLoss(**cfg['loss_config']['stft_config'])",
             "import inspect; inspect.signature(loss_fn).bind(net, (None, None), **cfg['loss_config'], mrstftloss=mr)",
             "loader = load_CleanNoisyPairDataset(root='synthetic:6', **cfg['trainset'], "
             "subset='training', batch_size=2, num_gpus=1); assert len(loader) == 3",
             "print('imports ok')"]
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "dropin"), ROOT]))
    out = subprocess.run([sys.executable, "-c", "\n".join(lines)], env=env, capture_output=True, text=True, timeout=300,
                         cwd=str(ROOT))
    assert out.returncode == 0 and "imports ok" in out.stdout, out.stderr[-3000:]


def test_pair_dataset_reads_crops_and_draws_like_the_reference(tmp_path):
    """dataset.py:301-390 host logic: directory layout, int16 -> [-1, 1) scaling, random crop, one random.choice per
    noise file and three per augmentation draw (in the reference's order), noise length check (D19)."""
    import random
    from scipy.io.wavfile import write as wavwrite
    from tinyrecurrentunet_amd import dataset as ds
    sr, crop = 16000, 0.25
    (tmp_path / "clean").mkdir()
    (tmp_path / "keyboard").mkdir()
    rng = np.random.default_rng(0)
    cleans = []
    for i in range(3):
        c = (rng.standard_normal(sr) * 3000).astype(np.int16)
        wavwrite(str(tmp_path / "clean" / ("fileid_%d.wav" % i)), sr, c)
        cleans.append(c.astype(np.float32) / 32768.0)
    noise = (rng.standard_normal(int(sr * crop)) * 2000).astype(np.int16)
    wavwrite(str(tmp_path / "keyboard" / "k0.wav"), sr, noise)
    d = ds.CleanNoisyPairDataset(root=str(tmp_path), subset="training", crop_length_sec=crop, sample_rate=sr)
    assert len(d) == 3
    random.seed(5)
    np.random.seed(5)
    clean, nz, fileid, params = d[1]
    assert fileid.endswith("fileid_1.wav") and clean.shape == (1, 4000) and nz.shape == (1, 4000)
    np.testing.assert_allclose(nz[0].numpy(), noise.astype(np.float32) / 32768.0)
    # the crop is a window of the file; the parameters are the draws the reference's DataAugment would make
    random.seed(5)
    np.random.seed(5)
    random.choice(d.noise_files)
    aug = ds.DataAugment()
    lp, hp, gain = aug.draw()
    start = np.random.randint(low=0, high=sr - 4000 + 1)
    np.testing.assert_array_equal(clean[0].numpy(), cleans[1][start:start + 4000])
    np.testing.assert_allclose(params.numpy(), aug.params(lp, hp, gain))
    assert 7000 <= lp < 10000 and 800 <= hp < 1200 and -12 <= gain < -5
    from oracle import augment_ref as ar
    b, a = ar.biquad_coeffs("lowpass", 48000, lp)
    np.testing.assert_allclose(params.numpy()[1:6], np.r_[b, a[1:]], rtol=1e-6)
    b, a = ar.biquad_coeffs("highpass", 48000, hp)
    np.testing.assert_allclose(params.numpy()[6:11], np.r_[b, a[1:]], rtol=1e-6)
    # a noise file of the wrong length is an error, as in the reference (clean[crop] + noise, dataset.py:380)
    wavwrite(str(tmp_path / "keyboard" / "k0.wav"), sr, noise[:100])
    with pytest.raises(ValueError):
        d[0]
    # the loader needs the GPU for the augmentation + mix: it fails loudly without one, never falls back
    ld = ds.load_CleanNoisyPairDataset(root="synthetic:4", subset="training", crop_length_sec=0.1, batch_size=2,
                                       sample_rate=sr, num_workers=0)
    if not torch.cuda.is_available():
        from tinyrecurrentunet_amd import _lib
        with pytest.raises(_lib.TrunetHipError):
            next(iter(ld))


def test_fused_adamw_state_dict_format():
    """optimizer_state_dict of train.py:157-161: FusedAdamW writes torch.optim.AdamW's structure and loads a state dict
    written by torch.optim.AdamW (host logic only: no step is taken without the GPU)."""
    from tinyrecurrentunet_amd import optim
    ps = [torch.nn.Parameter(torch.randn(4, 3)), torch.nn.Parameter(torch.randn(5))]
    ref = torch.optim.AdamW(ps, lr=4e-4)
    for p in ps:
        p.grad = torch.randn_like(p)
    ref.step()
    ref.step()
    sd = ref.state_dict()
    ours = optim.FusedAdamW(ps, lr=1e-3)
    ours.load_state_dict(sd)
    assert ours.param_groups[0]["lr"] == 4e-4
    back = ours.state_dict()
    assert set(back) == {"state", "param_groups"} and set(back["state"]) == {0, 1}
    assert set(back["param_groups"][0]) == set(sd["param_groups"][0])
    for i in (0, 1):
        assert float(back["state"][i]["step"]) == 2.0
        assert torch.equal(back["state"][i]["exp_avg"], sd["state"][i]["exp_avg"])
        assert torch.equal(back["state"][i]["exp_avg_sq"], sd["state"][i]["exp_avg_sq"])
    torch.optim.AdamW(ps, lr=4e-4).load_state_dict(back)           # and torch accepts what we write


def test_entry_points_reject_null_and_bad_shapes():
    """Every C entry point validates pointers and extents on the host before launching and returns TRUNET_EINVAL
    (no launch, no GPU needed): the guard behind round 1's unexplained NULL-dereference fault."""
    import ctypes as C
    from tinyrecurrentunet_amd import _lib
    lib = _lib.lib()
    EINVAL = _lib.TRUNET_EINVAL
    a = _lib.GemmArgs()
    a.NP, a.N, a.P, a.M, a.nseg, a.out_L, a.ldw_m, a.ldw_c = 256, 200, 4, 64, 1, 4, 64, 1
    a.seg[0] = _lib.Seg()
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # out / W / segment source all NULL
    a.out, a.W = 0x1000, 0x2000
    a.seg[0].src0, a.seg[0].nchan, a.seg[0].L, a.seg[0].pos_mul, a.seg[0].pos_div = 0x3000, 64, 4, 1, 1
    a.epi = _lib.EPI_MASK
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # EPI_MASK without zmask / e0 / e1
    a.zmask, a.e0, a.e1 = 0x4000, 0x5000, 0x6000
    a.epi = _lib.EPI_MASK | _lib.EPI_STATS
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # statistics without a partials buffer
    a.epi = _lib.EPI_BIAS
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # EPI_BIAS without bias
    a.epi = 0
    a.seg[0].mode = _lib.PRO_BNBWD
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # BatchNorm-backward prologue without src1 / coefficients
    a.seg[0].mode = _lib.PRO_BNRELU
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # BatchNorm prologue without c0 / c1
    a.seg[0].mode = _lib.PRO_NONE
    a.out_pos_off = 1
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # output rows p + 1 run past out_L
    a.out_pos_off, a.p_begin = -1, 0
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # ... or start before row 0
    a.out_pos_off = 0
    a.NP = 200
    assert lib.trunet_conv_gemm(a, None) == EINVAL                     # NP not a multiple of the frame tile
    w = _lib.WgradArgs()
    w.NP, w.N, w.P, w.M, w.nseg, w.a_L, w.ldw_m, w.ldw_c, w.w_numel = 256, 200, 4, 64, 1, 4, 64, 1, 4096
    w.seg[0] = a.seg[0]
    w.a0, w.w_partials = 0x1000, 0x2000
    w.a_mode = _lib.PRO_BNBWD
    assert lib.trunet_conv_wgrad(w, None) == EINVAL                    # BNBWD on dz without a1 / coefficients
    w.a_mode, w.a_pos_off = _lib.PRO_NONE, 2
    assert lib.trunet_conv_wgrad(w, None) == EINVAL                    # dz rows p + 2 run past a_L
    pb = _lib.PwBwdArgs()
    assert lib.trunet_pw_bwd(pb, None) == EINVAL
    assert lib.trunet_relu_bwd_stats(None, None, None, None, None, None, 4, 4, 256, 200, None) == EINVAL
    assert lib.trunet_augment_mix(None, None, None, None, None, 2, 100, None) == EINVAL
    assert lib.trunet_gru_fwd(None, None, None, None, None, None, None, 64, 16, 256, None) == EINVAL
    assert lib.trunet_gru_bwd(None, None, None, None, None, None, None, 64, 16, 256, 200, None) == EINVAL
    assert lib.trunet_stft_features(None, None, None, None, 1, 2048, 17, 3, None) == EINVAL
    assert lib.trunet_mask_istft_fwd(None, None, None, None, None, None, 1, 17, 2048, 0.5, None) == EINVAL
    assert lib.trunet_dwconv_fwd(None, None, None, None, None, None, None, 8, 3, 1, 4, 4, 256, 200, None) == EINVAL


def test_bf16_entry_points_reject_null_and_bad_shapes():
    """The trunet_bf16_* family validates on the host like the fp32 entry points (no launch, no GPU needed)."""
    import ctypes as C
    from tinyrecurrentunet_amd import _lib
    lib = _lib.lib()
    EINVAL, ENOTSUP = _lib.TRUNET_EINVAL, _lib.TRUNET_ENOTSUP
    a = _lib.BGemmArgs()
    a.NP, a.N, a.P, a.M, a.nseg, a.out_L, a.nks_total = 256, 200, 4, 64, 1, 4, 4
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # out / wfrag / segment source NULL
    a.out, a.wfrag = 0x1000, 0x2000
    sg = _lib.BSeg()
    sg.src0, sg.nchan, sg.L, sg.pos_mul, sg.pos_div = 0x3000, 64, 4, 1, 1
    a.seg[0] = sg
    a.epi = _lib.EPI_MASK
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # mask without zmask / e0 / e1
    a.epi = _lib.EPI_STATS
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # statistics without partials
    a.epi = _lib.EPI_BIAS
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # bias flag without bias
    a.epi = 0
    a.seg[0].mode = _lib.PRO_BNBWD
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # BatchNorm-backward pair without src1 / coefficients
    a.seg[0].mode = _lib.PRO_NONE
    a.out_pos_off = 1
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # output rows run past out_L
    a.out_pos_off = 0
    a.nks_total = 2
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # packed weight image shorter than the segments
    a.nks_total, a.NP = 4, 200
    assert lib.trunet_bf16_gemm(a, None) == EINVAL                     # NP not a multiple of 64
    a.NP, a.M = 256, 192
    assert lib.trunet_bf16_gemm(a, None) == ENOTSUP                    # more than 128 rows
    w = _lib.BWgradArgs()
    w.NP, w.N, w.P, w.M, w.nseg, w.a_L, w.ldw_m, w.ldw_c, w.w_numel = 256, 200, 4, 64, 1, 4, 64, 1, 4096
    w.seg[0] = sg
    w.a0, w.w_partials = 0x1000, 0x2000
    w.a_mode = _lib.PRO_BNBWD
    assert lib.trunet_bf16_wgrad(w, None) == EINVAL                    # BatchNorm backward on dz without z / coefficients
    w.a_mode, w.a_pos_off = _lib.PRO_NONE, 2
    assert lib.trunet_bf16_wgrad(w, None) == EINVAL                    # dz rows run past a_L
    w.a_pos_off = 0
    w.seg[0].pos_div = 3
    assert lib.trunet_bf16_wgrad(w, None) == ENOTSUP                   # positions are shifts: strides 1 and 2 only
    w.seg[0].pos_div = 1
    w.seg[0].nchan = 400
    assert lib.trunet_bf16_wgrad(w, None) == ENOTSUP                   # more source octets than the LDS image holds
    n3 = (C.c_int32 * 1)(64)
    assert lib.trunet_bf16_pack_weight(None, 0x1000, 64, 64, 1, 0, 1, n3, n3, None) == EINVAL
    assert lib.trunet_bf16_dwconv_fwd(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 12, 3, 1, 8, 8, 256, 200,
                                      None) == EINVAL                   # channels not a multiple of 8
    assert lib.trunet_bf16_dwconv_fwd(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 16, 3, 1, 8, 7, 256, 200,
                                      None) == EINVAL                   # Lout inconsistent with (Lin, K, S)
    assert lib.trunet_bf16_dwconv_fwd(0x1000, 0x1000, 0x1000, 0x1000, 0x1000, 0x1000, None, 16, 3, 1, 8, 8, 256, 200,
                                      None) == EINVAL                   # no partials buffer
    assert lib.trunet_bf16_from_frames_last(None, 0x1000, 8, 4, 256, None) == EINVAL
    # round-3 entry points: argument checks come before any launch
    P = 0x1000
    assert lib.trunet_bf16_gru_fwd(None, P, P, P, P, P, P, 64, 16, 256, None) == EINVAL
    assert lib.trunet_bf16_gru_fwd(P, P, P, P, P, P, None, 64, 16, 200, None) == EINVAL        # NP not a multiple of 128
    assert lib.trunet_bf16_gru_fwd(P, P, P, P, P, P, None, 32, 16, 256, None) == ENOTSUP       # hidden size other than 64
    assert lib.trunet_bf16_gru_bwd(P, P, None, P, P, P, P, 64, 16, 256, None) == EINVAL         # backward needs the saved gates
    assert lib.trunet_bf16_gru_bwd(P, P, P, P, P, P, P, 128, 16, 256, None) == ENOTSUP
    assert lib.trunet_dwconv_bwd_rz(P, None, P, P, P, P, P, P, P, P, P, P, P, P, 8, 3, 1, 64, 64, 256, 200, None) == EINVAL
    assert lib.trunet_dwconv_bwd_rz(P, P, P, P, P, P, P, P, P, P, P, P, P, P, 8, 3, 1, 64, 64, 200, 200, None) == EINVAL  # NP % 128
    assert lib.trunet_dwconv_bwd_rz(P, P, P, P, P, P, P, P, P, P, P, P, P, P, 8, 7, 1, 64, 64, 256, 200, None) == ENOTSUP  # k = 7
    assert lib.trunet_dwconv_bwd_rz(P, P, P, P, P, P, P, P, P, P, P, P, P, P, 8, 3, 2, 32, 16, 256, 200, None) == ENOTSUP  # < 32 outputs
    assert lib.trunet_bf16_dw_nparts(512, 64) == 2 * 4 and lib.trunet_bf16_gemm_nparts() > 0
    # fp32 GEMM family: strides other than 1 / 2 are refused (segment positions are computed with shifts)
    g = _lib.GemmArgs()
    g.NP, g.N, g.P, g.M, g.nseg, g.out_L, g.ldw_m, g.ldw_c = 256, 200, 4, 64, 1, 4, 64, 1
    g.out, g.W = 0x1000, 0x2000
    g.seg[0] = _lib.Seg()
    g.seg[0].src0, g.seg[0].nchan, g.seg[0].L, g.seg[0].pos_mul, g.seg[0].pos_div = 0x3000, 64, 4, 1, 3
    assert lib.trunet_conv_gemm(g, None) == ENOTSUP


def test_bf16_kernel_name_mirror_and_precision_switch():
    """engine_bf16._bgemm_name mirrors trunet_bf16_gemm's dispatch (bench.py matches it against rocprofv3 names);
    TRUNet(precision=...) validates its argument without touching the GPU"""
    from tinyrecurrentunet_amd import _lib
    from tinyrecurrentunet_amd.engine_bf16 import _bgemm_name
    from tinyrecurrentunet_amd.network import TRUNet

    class S:
        def __init__(self, nchan, mode):
            self.nchan, self.mode = nchan, mode
    B, St, A, K = _lib.EPI_BIAS, _lib.EPI_STATS, _lib.EPI_ACCUM, _lib.EPI_MASK
    assert _bgemm_name(128, [S(128, _lib.PRO_BNRELU)], B | St) == "bgemm_kernel<4, 1, 3, true>"
    assert _bgemm_name(64, [S(64, _lib.PRO_BNRELU), S(128, _lib.PRO_NONE)], B | St) == "bgemm_kernel<2, 1, 3, true>"
    assert _bgemm_name(128, [S(64, _lib.PRO_BNBWD)], K | St | A) == "bgemm_kernel<2, 2, 14, true>"
    assert _bgemm_name(64, [S(4, _lib.PRO_NONE)] * 5, B | _lib.EPI_RELU) == "bgemm_kernel<2, 0, 17, false>"
    assert _bgemm_name(8, [S(64, _lib.PRO_BNRELU)], B | St) == "bgemm_kernel<1, 1, 3, false>"
    assert _bgemm_name(8, [S(64, _lib.PRO_BNRELU)], B | St | A) == "bgemm_kernel<1, -1, -1, false>"
    net = TRUNet(input_size=4, precision="bf16")
    assert net.precision == "bf16" and len(net.state_dict()) == 177
    assert net.set_precision("fp32").precision == "fp32"
    with pytest.raises(ValueError):
        TRUNet(precision="fp16")


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


def test_bare_bench_launcher_fails_loudly_when_a_rank_dies():
    """`python bench.py --gpus 2` is its own launcher (reference: distributed.py:150-176).  Without a GPU no rank can
    run, so what is checked here is the launcher: it starts the ranks without importing the product or touching the
    GPU itself, a rank that dies takes the job down (the other rank is terminated, not left waiting at the rendezvous),
    and the parent's exit code is non-zero with no JSON line on stdout."""
    import subprocess
    env = dict(os.environ, TRUNET_BENCH_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "stopping the other ranks" in out.stderr or "exited with code" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_bench_ranks_pin_themselves_to_disjoint_core_slices():
    """every bench rank restricts itself to its own slice of the node's cores before torch is imported
    (under torch.distributed.run and under bench.py's own launcher alike); --host-cores K narrows a rank to K cores."""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    cores = list(range(3, 35))                          # 32 cores with an offset
    sl = [bench.rank_core_slice(cores, r, 8) for r in range(8)]
    assert all(len(x) == 4 for x in sl) and sorted(sum(sl, [])) == cores
    assert bench.rank_core_slice(cores, 0, 1) == cores
    assert bench.rank_core_slice([5, 9], 3, 8) == [9] and bench.rank_core_slice([5, 9], 2, 8) == [5]
    sl = [bench.rank_core_slice(list(range(10)), r, 4) for r in range(4)]       # 10 cores, 4 ranks: 2 each, disjoint
    assert all(len(x) == 2 for x in sl) and len(set(sum(sl, []))) == 8
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 4:
        pytest.skip("needs 4 host cores")
    code = ("import os, sys; sys.argv = ['bench.py'] + sys.argv[1:]; sys.path.insert(0, %r); import bench; "
            "print(bench.pin_host_cores(), sorted(os.sched_getaffinity(0)))" % ROOT)
    for env, argv, want in (({"LOCAL_RANK": "1", "LOCAL_WORLD_SIZE": "2"}, [], have[len(have) // 2:2 * (len(have) // 2)]),
                            ({"LOCAL_RANK": "0", "LOCAL_WORLD_SIZE": "2"}, ["--host-cores", "1"], have[:1]),
                            ({}, ["--host-cores=2"], have[:2]),
                            ({"TRUNET_BENCH_PIN": "0", "LOCAL_RANK": "1", "LOCAL_WORLD_SIZE": "2"}, [], None)):
        e = {k: v for k, v in os.environ.items() if k not in ("LOCAL_RANK", "LOCAL_WORLD_SIZE", "WORLD_SIZE", "TRUNET_BENCH_PIN")}
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code] + argv, env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        assert out.stdout.strip() == "%s %s" % (want, want if want is not None else have), (env, argv, out.stdout)


def test_use_tgru_bf16_is_refused_at_construction():
    from tinyrecurrentunet_amd import _lib, network as hn
    with pytest.raises(_lib.TrunetHipError):
        hn.TRUNet(input_size=4, use_tgru=True, precision="bf16")
