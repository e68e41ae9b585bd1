"""Drop-in shim: `import phm` (as /root/reference/train.py, rt.py, ... do) resolves to the MI355X-native module."""
from tinyrecurrentunet_amd.phm import *  # noqa: F401,F403
from tinyrecurrentunet_amd import phm as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("__")]
globals().update({n: getattr(_impl, n) for n in dir(_impl) if not n.startswith("__")})
