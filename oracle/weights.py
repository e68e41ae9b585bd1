"""Deterministic weights / BatchNorm statistics for parity tests.  TEST INFRASTRUCTURE.

numpy PCG64 streams keyed by (seed, tensor name), so the GPU box rebuilds identical
tensors without shipping them; fixtures store a checksum to detect RNG drift.
Scales follow torch's default inits (U(-1/sqrt(fan_in), +)); BN affine and running
statistics are randomised so that eval-mode folding and training-mode statistics are
both exercised non-trivially.
"""
import zlib

import numpy as np
import torch


def _rng(seed, name):
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def fill_state_dict(module, seed=0, dtype=torch.float32):
    """Overwrite every entry of ``module.state_dict()`` in place, deterministically."""
    sd = module.state_dict()
    new = {}
    for name, t in sd.items():
        g = _rng(seed, name)
        shape = tuple(t.shape)
        if name.endswith("num_batches_tracked"):
            new[name] = torch.zeros((), dtype=torch.long)
            continue
        if name.endswith("running_mean"):
            v = g.normal(0.0, 0.2, shape)
        elif name.endswith("running_var"):
            v = g.uniform(0.5, 1.5, shape)
        elif t.dim() == 1 and ("." + name).rsplit(".", 2)[-2].isdigit() and _is_bn(sd, name):
            v = g.uniform(0.6, 1.4, shape) if name.endswith("weight") else g.normal(0.0, 0.2, shape)
        else:
            fan_in = int(np.prod(shape[1:])) if t.dim() > 1 else None
            if fan_in is None:                       # bias: use sibling weight's fan-in
                w = sd.get(name.replace("bias", "weight"), None)
                fan_in = int(np.prod(tuple(w.shape)[1:])) if w is not None and w.dim() > 1 else shape[0]
                if "GRU" in name:
                    fan_in = 64 if "FGRU" in name else 128
            elif "GRU.weight" in name:
                fan_in = 64 if "FGRU" in name else 128
            b = 1.0 / np.sqrt(max(fan_in, 1))
            v = g.uniform(-b, b, shape)
        new[name] = torch.tensor(v, dtype=dtype)
    module.load_state_dict(new)
    return module


def _is_bn(sd, name):
    return name.rsplit(".", 1)[0] + ".running_mean" in sd


def checksum(module):
    """Order-independent fp64 checksum of all floating tensors of the state dict."""
    tot = 0.0
    for name, t in module.state_dict().items():
        if t.is_floating_point():
            tot += float(t.double().abs().sum()) * (1 + (zlib.crc32(name.encode()) % 97) / 97.0)
    return tot


def synth_pairs(B, L, seed=1234, dtype=torch.float32):
    """Synthetic DNS-shaped pairs: clean = 2-tap-mean low-passed 0.1*randn,
    noisy = clean + 0.05*randn.  Returns (clean, noisy), each (B,1,L)."""
    g = np.random.default_rng(seed)
    c = 0.1 * g.standard_normal((B, 1, L + 1))
    clean = 0.5 * (c[..., 1:] + c[..., :-1])
    noisy = clean + 0.05 * g.standard_normal((B, 1, L))
    return torch.tensor(clean, dtype=dtype), torch.tensor(noisy, dtype=dtype)
