"""tinyrecurrentunet_amd -- MI355X-native hot path of Tiny Recurrent U-Net (HIP kernels behind a C ABI).

Modules mirror the reference's own (``network``, ``dataset``, ``phm``, ``stft_loss``, ``distributed``,
``util``); see INTEGRATION.md for the drop-in shims.
"""
__all__ = ["network"]
