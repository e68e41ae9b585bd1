"""MI355X-native ``network`` module: same classes, constructor signatures and ``state_dict`` keys
as ``/root/reference/network.py`` (blocks :9-120, ``TRUNet`` :122-171 in the repaired composition
R1-R4 of SURVEY.md section 0.2), with every forward/backward running on the hand-written HIP
kernels of libtrunet_hip.so.  The ``torch.nn`` layers below are parameter containers only (they
give the reference's parameter names, shapes and default initialisation); their own ``forward``
is never called.  There is no CPU path: calling a module with a CPU tensor raises.
"""
import torch
import torch.nn as nn

from . import _lib
from .engine import TRUNetEngine


def _need_gpu(x):
    if not x.is_cuda:
        raise _lib.TrunetHipError("tinyrecurrentunet_amd runs on MI355X only: got a %s tensor "
                                  "(the CPU restatement lives in oracle/, for tests)" % x.device)
    _lib.lib()


def _pw_bn_relu(cin, cout):
    return [nn.Conv1d(cin, cout, kernel_size=1), nn.BatchNorm1d(cout), nn.ReLU(inplace=True)]


class _BlockFn(torch.autograd.Function):
    """One autograd node per stand-alone block: forward/backward are the same HIP launches the fused ``TRUNet``
    schedule uses for that block (engine.block_forward / block_backward); gradients for parameters and inputs."""

    @staticmethod
    def forward(ctx, engine, run, training, nx, *tensors):
        out, ectx = run(training)          # eval-mode forwards record nothing (their backward raises)
        ctx.engine, ctx.ectx, ctx.nx, ctx.training = engine, ectx, nx, training
        ctx.params = tensors[nx:]
        return out

    @staticmethod
    def backward(ctx, gout):
        if not ctx.training:
            raise _lib.TrunetHipError("backward through a block in eval() mode is not supported (BatchNorm running "
                                      "statistics); call .train()")
        grads, gxs = ctx.engine.block_backward(ctx.ectx, gout)
        return (None, None, None, None) + tuple(gxs) + tuple(grads.get(p) for p in ctx.params)


def _run_block(mod, run, xs):
    """run(record) -> (y, ctx).  With autograd recording: through _BlockFn (backward in training mode only, like TRUNet)."""
    if torch.is_grad_enabled() and (any(p.requires_grad for p in mod.parameters()) or any(x.requires_grad for x in xs)):
        return _BlockFn.apply(mod._engine, run, mod.training, len(xs), *xs, *mod.parameters())
    return run(False)[0]


class _Block(nn.Module):
    """Stand-alone use of a block class (the reference's own forward signatures and autograd behaviour)."""
    _kind = None
    _seq = None

    def _run(self, *xs):
        for x in xs:
            _need_gpu(x)
        if getattr(self, "_engine", None) is None:
            object.__setattr__(self, "_engine", TRUNetEngine(self))
        xf = [x.float() for x in xs]
        return _run_block(self, lambda rec: self._engine.block_forward(self._kind, getattr(self, self._seq), xf,
                                                                      self.training, record=rec), xs)


class StandardConv1d(_Block):
    """network.py:9-21."""
    _kind, _seq = "std", "StandardConv1d"

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.StandardConv1d = nn.Sequential(
            nn.Conv1d(in_channels, out_channels, kernel_size, stride=stride, padding=stride // 2),
            nn.ReLU(inplace=True))

    def forward(self, x):
        return self._run(x)


class DepthwiseSeparableConv1d(_Block):
    """network.py:24-43."""
    _kind, _seq = "dsc", "DepthwiseSeparableConv1d"

    def forward(self, x):
        return self._run(x)

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.DepthwiseSeparableConv1d = nn.Sequential(
            *_pw_bn_relu(in_channels, out_channels),
            nn.Conv1d(out_channels, out_channels, kernel_size, stride=stride, padding=kernel_size // 2,
                      groups=out_channels),
            nn.BatchNorm1d(out_channels), nn.ReLU(inplace=True))


class GRUBlock(nn.Module):
    """network.py:45-58."""

    def __init__(self, in_channels, hidden_size, out_channels, bidirectional):
        super().__init__()
        self.GRU = nn.GRU(in_channels, hidden_size, batch_first=True, bidirectional=bidirectional)
        width = hidden_size * (2 if bidirectional else 1)
        self.conv = nn.Sequential(*_pw_bn_relu(width, out_channels))

    def forward(self, x):
        _need_gpu(x)
        if getattr(self, "_engine", None) is None:
            object.__setattr__(self, "_engine", TRUNetEngine(self))
        xf = x.float()
        return _run_block(self, lambda rec: self._engine.gru_block_forward(self, xf, self.training, record=rec), [x])


def _trcnn_body(in_channels, out_channels, kernel_size, stride, tail=True):
    layers = _pw_bn_relu(in_channels, out_channels)
    layers.append(nn.ConvTranspose1d(out_channels, out_channels, kernel_size, stride=stride, padding=stride // 2))
    if tail:
        layers += [nn.BatchNorm1d(out_channels), nn.ReLU(inplace=True)]
    return nn.Sequential(*layers)


class FirstTrCNN(_Block):
    """network.py:60-76."""
    _kind, _seq = "first_tr", "FirstTrCNN"

    def forward(self, x):
        return self._run(x)

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.FirstTrCNN = _trcnn_body(in_channels, out_channels, kernel_size, stride)


class TrCNN(_Block):
    """network.py:79-100."""
    _kind, _seq = "tr", "TrCNN"

    def forward(self, x1, x2):
        return self._run(x1, x2)

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.TrCNN = _trcnn_body(in_channels, out_channels, kernel_size, stride)


class LastTrCNN(_Block):
    """network.py:102-120."""
    _kind, _seq = "last_tr", "LastTrCNN"

    def forward(self, x1, x2):
        return self._run(x1, x2)

    def __init__(self, in_channels, out_channels, kernel_size, stride):
        super().__init__()
        self.LastTrCNN = _trcnn_body(in_channels, out_channels, kernel_size, stride, tail=False)


class _TRUNetFn(torch.autograd.Function):
    """One autograd node for the whole body: forward and backward are the HIP schedules of
    ``engine.TRUNetEngine``; gradients are produced for the parameters only (the input features
    carry no gradient in the reference either, util.py:214-218)."""

    @staticmethod
    def forward(ctx, x, engine, training, tgru_T, *params):
        out, ectx = engine.forward(x, training, tgru_T=tgru_T, record=training)
        ctx.engine, ctx.ectx, ctx.params = engine, ectx, params
        ctx.training = training
        return out

    @staticmethod
    def backward(ctx, gout):
        if not ctx.training:
            raise _lib.TrunetHipError("backward through TRUNet in eval() mode is not supported "
                                      "(BatchNorm running statistics); call net.train()")
        grads = ctx.engine.backward(ctx.ectx, gout)
        return (None, None, None, None) + tuple(grads.get(p) for p in ctx.params)


class TRUNet(nn.Module):
    """Tiny Recurrent U-Net body: (N, C_in, 257) -> (N, 8, 257), N = frames.

    Constructor arguments as in network.py:123-130; like the reference only the layer sizes of
    network.py:134-150 exist, and ``input_size`` sets the first conv's channels (R2).  ``TGRU`` is
    registered for checkpoint parity and not executed (R4)."""

    def __init__(self, input_size=3, channels_input=64, channels_output=3, channels_hidden=128,
                 kernel_sizes=(5, 3), strides=(2, 1), tr_channels_input=192, use_tgru=False, precision="fp32"):
        super().__init__()
        # precision (extension; BASELINE.json configs[2]): "bf16" stores activations and their gradients as bf16 and
        # multiplies on the bf16 MFMA (engine_bf16.py); parameters, BatchNorm statistics, gradients of parameters stay fp32
        # use_tgru (extension, default = the reference as written, R4): run the TGRU block over time between FGRU and
        # the decoder as drawn in docs/net.jpg; forward then needs frames_per_seq = T (N = B*T frames)
        self.use_tgru = bool(use_tgru)
        self.precision = "fp32"
        self.set_precision(precision)      # after use_tgru: the combination (use_tgru, "bf16") is refused here
        self.encoder = nn.ModuleList([
            StandardConv1d(input_size, 64, 5, 2),
            DepthwiseSeparableConv1d(64, 128, 3, 1),
            DepthwiseSeparableConv1d(128, 128, 5, 2),
            DepthwiseSeparableConv1d(128, 128, 3, 1),
            DepthwiseSeparableConv1d(128, 128, 5, 2),
            DepthwiseSeparableConv1d(128, 128, 3, 2)])
        self.decoder = nn.ModuleList([
            FirstTrCNN(64, 64, 3, 2),
            TrCNN(192, 64, 5, 2),
            TrCNN(192, 64, 3, 1),
            TrCNN(192, 64, 5, 2),
            TrCNN(192, 64, 3, 1),
            LastTrCNN(128, 8, 5, 2)])
        self.FGRU = GRUBlock(128, 64, 64, bidirectional=True)
        self.TGRU = GRUBlock(64, 128, 64, bidirectional=False)
        object.__setattr__(self, "_engine", None)

    def set_precision(self, precision):
        """"fp32" (the reference's precision, default) or "bf16" (activation storage + MFMA operands; training forward /
        backward of the full network only -- eval forwards take the folded fp32 single-launch path as before)."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (precision,))
        if precision == "bf16" and self.__dict__.get("use_tgru", False):
            raise _lib.TrunetHipError("use_tgru is fp32 only")
        if precision != self.precision:
            object.__setattr__(self, "_engine", None)
        self.precision = precision
        return self

    def train(self, mode=True):
        if mode:
            object.__setattr__(self, "_folded_cache", None)      # training moves weights and BatchNorm statistics
        return super().train(mode)

    def _make_engine(self):
        if self.precision == "bf16":
            from .engine_bf16 import TRUNetEngineBF16
            return TRUNetEngineBF16(self)
        return TRUNetEngine(self)

    def stream_step(self, x, state=None):
        """Stateful causal streaming with the time-recurrent block: ``x`` (streams, C_in, 257) is ONE
        new STFT frame per stream; the encoder, FGRU and decoder run as in ``forward`` (eval mode) and the TGRU block
        (network.py:150, constructed but never called by the reference, D6) runs one GRU time step per (stream,
        frequency position) between FGRU and the decoder, as drawn in docs/net.jpg.  Returns (y, state); pass the
        state back in with the next frame.  Inference only."""
        _need_gpu(x)
        if self.training:
            raise _lib.TrunetHipError("stream_step is an inference path: call net.eval() first")
        if self.precision != "fp32":
            raise _lib.TrunetHipError("stream_step is fp32 only")
        if state is None:
            state = TRUNetStreamState()
        if state.layout is None:
            state.layout = "folded" if (self.fold_eval and 0 < x.shape[0] <= self.fold_max_frames) else "frames_last"
        if state.layout == "folded":
            # the whole step -- encoder, FGRU, ONE TGRU time step per (stream, frequency position), decoder -- in ONE launch
            # (export.fold(tgru=True) + stream_fwd_kernel<true>); the state (streams, 128, 16) is updated in place
            run = self.folded(tgru=True)
            if state.h is None:
                state.h, state.n = run.new_state(x.shape[0], x.device), x.shape[0]
            if state.n != x.shape[0]:
                raise _lib.TrunetHipError("stream state belongs to %d streams, got %d" % (state.n, x.shape[0]))
            with torch.no_grad():
                out = run.stream_step(x, state.h)
            state.steps += 1
            return out, state
        if self._engine is None:
            object.__setattr__(self, "_engine", TRUNetEngine(self))
        with torch.no_grad():
            out, _ = self._engine.forward(x.float(), False, tgru_state=state)
        return out, state

    def stream_audio(self, chunk, state=None, tgru=None):
        """Causal audio-in -> audio-out step (stream.py:83-109): ``chunk`` (streams, 128) new samples per stream ->
        (denoised samples that became final, state).  ``state`` is a ``streaming.AudioStream`` (created on the first call:
        analysis ring, PCEN smoother, overlap-add tail and -- ``tgru`` / ``use_tgru`` -- the TGRU hidden state); call
        ``state.flush()`` at the end of the utterance.  Three launches per hop; see streaming.py for the latency contract."""
        _need_gpu(chunk)
        if state is None:
            from .streaming import AudioStream
            state = AudioStream(self, chunk.shape[0], tgru=self.use_tgru if tgru is None else tgru)
        return state.push(chunk), state

    def _active_params(self):
        return [p for n, p in self.named_parameters() if self.use_tgru or not n.startswith("TGRU.")]

    def forward(self, x, frames_per_seq=None):
        _need_gpu(x)
        if self._engine is None:
            object.__setattr__(self, "_engine", self._make_engine())
        T = None
        if self.use_tgru:
            if frames_per_seq is None:
                raise _lib.TrunetHipError("use_tgru: pass frames_per_seq = T (x holds B utterances of T frames each)")
            T = int(frames_per_seq)
        params = self._active_params()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            return _TRUNetFn.apply(x, self._engine, self.training, T, *params)
        if not self.training and T is None and self.fold_eval and 0 < x.shape[0] <= self.fold_max_frames:
            # eval, no autograd: BatchNorm folded into the convs, the whole forward in ONE launch (export.py,
            # stream_fwd.hip) -- the streaming / rt.py-protocol path
            return self.folded()(x)
        out, _ = self._engine.forward(x, self.training, tgru_T=T)
        return out

    # eval-mode single-launch forward: on by default for batches up to fold_max_frames frames (beyond that the
    # layer-by-layer kernels, which share one weight load over all frames, are faster)
    fold_eval = True
    fold_max_frames = 8192

    # Weights can also be written behind torch's back: `p.data.mul_()` / `layer.weight.data /= s` (the idiom of the reference's
    # util.weight_scaling_init, util.py:168-175) move neither the parameter's version counter nor the mutation epoch.  With
    # fold_verify (default) every eval call that may reuse a cached artefact first takes a 64-bit content checksum of all
    # parameters and buffers on the device (one launch + one 8-byte read-back, ~50 us) and re-folds when it moved.  Serving
    # loops with frozen weights can switch it off (net.fold_verify = False) and call net.invalidate_folded() after a raw write.
    # During hipGraph capture nothing can be read back: the artefact captured is the one verified by the warm-up call.
    fold_verify = True

    def invalidate_folded(self):
        """Drop the cached eval artefacts (rebuilt by the next eval forward / stream_step)."""
        object.__setattr__(self, "_folded_cache", None)

    def _content_checksum(self, ts):
        import ctypes as C
        ptrs = tuple((t.data_ptr(), t.numel() * t.element_size() // 4) for t in ts)
        ck = self.__dict__.get("_cksum")
        if ck is None or ck[0] != ptrs or ck[1].device != ts[0].device:
            arr = (C.c_int64 * (2 * len(ptrs)))(*[v for pn in ptrs for v in pn])
            desc = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.int64).to(ts[0].device)
            ck = (ptrs, desc, ts)                           # ts: keeps the described storages alive
            object.__setattr__(self, "_cksum", ck)
        out = torch.zeros(1, device=ts[0].device, dtype=torch.int64)
        _lib.check(_lib.lib().trunet_checksum_batch(ck[1].data_ptr(), len(ptrs), out.data_ptr(), _lib.stream()), "checksum")
        return int(out.item())

    def folded(self, tgru=False):
        """The exported inference artefact of the current weights (export.FoldedTRUNet), rebuilt when a parameter or
        buffer has been modified since; tgru=True: with the time-recurrent block (the stateful stream_step artefact)."""
        from .export import FoldedTRUNet
        ts = [t for n, t in self.state_dict(keep_vars=True).items() if tgru or not n.startswith("TGRU.")]
        # version counters catch torch-side writes (load_state_dict, in-place ops); the mutation epoch catches the
        # product's own raw-pointer writers (FusedAdamW.step, the BatchNorm running statistics of a training forward);
        # the data pointers catch re-pointed storages; the content checksum everything else (see fold_verify)
        cache = self.__dict__.get("_folded_cache")
        if cache is None:
            cache = {}
            object.__setattr__(self, "_folded_cache", cache)
        cached = cache.get(bool(tgru))
        key = (tuple(t._version for t in ts), tuple(t.data_ptr() for t in ts), str(ts[0].device), _lib.mutation_epoch())
        if cached is not None and cached[0][:4] == key and (not self.fold_verify or torch.cuda.is_current_stream_capturing()):
            return cached[1]
        if self.fold_verify and not torch.cuda.is_current_stream_capturing():
            key = key + (self._content_checksum(ts),)
        if cached is None or cached[0] != key:
            cached = (key, FoldedTRUNet.from_module(self, tgru=tgru))
            cache[bool(tgru)] = cached
        return cached[1]


class TRUNetStreamState:
    """Hidden state of the TGRU block for ``TRUNet.stream_step``: one 128-vector per (stream, frequency position).
    layout "folded" (the single-launch path): h (streams, 128, 16); layout "frames_last" (the layer-by-layer kernels, more
    than ``fold_max_frames`` streams or ``fold_eval = False``): h [128][16][NP], one column per stream."""

    def __init__(self, layout=None):
        self.h, self.n, self.steps, self.layout = None, 0, 0, layout

    def hidden(self):
        """(streams, 16, 128) view of the state, whatever the layout"""
        if self.h is None:
            return None
        if self.layout == "folded":
            return self.h.permute(0, 2, 1)
        return self.h[:, :, :self.n].permute(2, 1, 0)


# train.py:22 imports this name (it does not exist in the reference either): alias only.
TRUNet2D = TRUNet
