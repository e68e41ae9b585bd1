"""Drop-in shim: `import stft_loss` (as /root/reference/train.py, rt.py, ... do) resolves to the MI355X-native module."""
from tinyrecurrentunet_amd.stft_loss import *  # noqa: F401,F403
from tinyrecurrentunet_amd import stft_loss as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in dir(_impl) if not n.startswith("__")})
