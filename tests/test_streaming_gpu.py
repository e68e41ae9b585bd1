"""GPU: the causal audio-in -> audio-out stream (tinyrecurrentunet_amd/streaming.py; stream.py:83-109 protocol) against the
offline path of util.loss_fn and against the oracle: the concatenated hops must BE the offline denoised audio."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def _pair(cin, seed, use_tgru=False):
    from oracle import network_ref as nr, weights as W
    from tinyrecurrentunet_amd import network as hn
    ref = W.fill_state_dict(nr.TRUNet(input_size=cin), seed=seed)
    net = hn.TRUNet(input_size=cin, use_tgru=use_tgru)
    net.load_state_dict(ref.state_dict())
    return ref, net.cuda().eval()


def _stream_all(net, x, tgru=False):
    S, Ln = x.shape
    outs, state = [], None
    lens = []
    for k in range(Ln // 128):
        o, state = net.stream_audio(x[:, 128 * k:128 * (k + 1)].contiguous(), state, tgru=tgru)
        lens.append(o.shape[1])
        outs.append(o)
    outs.append(state.flush())
    return torch.cat(outs, 1), lens, state


@pytest.mark.parametrize("cin,S,hops", [(4, 3, 37), (3, 2, 12), (4, 5, 3), (4, 1, 4)])
def test_audio_stream_equals_the_offline_path_and_the_oracle(cin, S, hops):
    """features (+ PCEN with the carried smoother) -> folded network -> mask -> inverse STFT with a carried overlap-add tail,
    one hop at a time, against (a) the offline HIP path on the whole utterance (stft_features / pcen -> eval forward ->
    util.denoise): the same arithmetic per frame, so the stream must reproduce it to rounding EVERYWHERE, the reflect-padded
    edges included, and (b) the oracle composition on the host (features_ref -> oracle network in eval mode ->
    denoise_from_output) at 1e-4; latency contract: nothing for three hops, then two hops, then one per hop, flush = rest."""
    from oracle import features_ref as fr
    from tinyrecurrentunet_amd import dataset as ds, util
    ref, net = _pair(cin, seed=3)
    Ln = 128 * hops
    x = torch.tensor(np.random.default_rng(hops).standard_normal((S, Ln)) * 0.1, dtype=torch.float32)
    xg = x.cuda()
    T = 1 + Ln // 128
    with torch.no_grad():
        feats = ds.stft_features(xg, pcen=(cin == 4))
        y = net(feats)
        off, _ = util.denoise(y, torch.zeros_like(xg), T)
    got, lens, state = _stream_all(net, xg)
    assert got.shape == (S, Ln)
    assert lens[:2] == [0, 0] and lens[2] == 0 and all(v == 128 for v in lens[4:]), lens
    assert state.frames == T
    assert _rel(got, off) < 1e-5, _rel(got, off)
    ref.eval()
    with torch.no_grad():
        fo = fr.features_batch(x[:, None, :], pcen=(cin == 4))
        den = fr.denoise_from_output(ref(fo), T, 0.5, length=Ln)
    assert _rel(got, den) < 1e-4, _rel(got, den)
    with pytest.raises(Exception):
        state.push(xg[:, :128].contiguous())          # a flushed stream takes no more audio


def test_audio_stream_with_the_time_recurrent_block_and_graph_replay():
    """tgru=True: the TGRU hidden state rides along (stream_fwd_kernel<true>); offline reference = the use_tgru network on
    the whole utterance (layer-by-layer kernels, frames_per_seq = T).  Then the same stream with its steady-state hop
    captured ONCE as a hipGraph and replayed hop after hop (three launches, all state at fixed addresses)."""
    from tinyrecurrentunet_amd import dataset as ds, util
    from tinyrecurrentunet_amd.streaming import AudioStream
    _, net = _pair(4, seed=5, use_tgru=True)
    S, hops = 4, 21
    Ln = 128 * hops
    xg = torch.tensor(np.random.default_rng(1).standard_normal((S, Ln)) * 0.1, dtype=torch.float32).cuda()
    T = 1 + Ln // 128
    with torch.no_grad():
        feats = ds.stft_features(xg, pcen=True)
        y = net(feats, frames_per_seq=T)
        off, _ = util.denoise(y, torch.zeros_like(xg), T)
    got, _, _ = _stream_all(net, xg, tgru=True)
    assert _rel(got, off) < 1e-4, _rel(got, off)
    # ---- graph replay of the steady-state hop
    st = AudioStream(net, S, tgru=True)
    outs = [st.push(xg[:, 128 * k:128 * (k + 1)].contiguous()) for k in range(6)]
    buf = xg[:, :128].clone()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    buf.copy_(xg[:, 128 * 6:128 * 7])
    with torch.cuda.graph(g):
        o = st._frame(buf)
    st.hops_in += 1
    g.replay()                                   # (capture does not execute)
    outs.append(o.clone())
    for k in range(7, hops):
        buf.copy_(xg[:, 128 * k:128 * (k + 1)])
        g.replay()
        outs.append(o.clone())
        st.hops_in += 1
        st.frames += 1
    outs.append(st.flush())
    rep = torch.cat(outs, 1)
    assert rep.shape == (S, Ln) and torch.equal(rep, got)
