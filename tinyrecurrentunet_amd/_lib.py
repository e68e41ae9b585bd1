"""ctypes binding of libtrunet_hip.so (the C ABI declared in include/trunet_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails this raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TRUNET_HIP_LIB: another build of the same library (diagnostic / A-B builds under scripts/dbg); default: the in-tree one
LIB_PATH = os.environ.get("TRUNET_HIP_LIB") or os.path.join(_HERE, "csrc", "libtrunet_hip.so")

MAX_SEG = 5
TRUNET_OK, TRUNET_EINVAL, TRUNET_ELAUNCH, TRUNET_ENOTSUP = 0, -1, -2, -3
PRO_NONE, PRO_BNRELU, PRO_BNBWD = 0, 1, 2
EPI_BIAS, EPI_STATS, EPI_ACCUM, EPI_MASK, EPI_RELU, EPI_F32OUT, EPI_PREZERO = 1, 2, 4, 8, 16, 32, 64

_fp = C.c_void_p


class Seg(C.Structure):
    _fields_ = [("src0", _fp), ("src1", _fp), ("c0", _fp), ("c1", _fp), ("c2", _fp),
                ("nchan", C.c_int32), ("L", C.c_int32), ("pos_mul", C.c_int32), ("pos_off", C.c_int32),
                ("pos_div", C.c_int32), ("woff", C.c_int32), ("mode", C.c_int32), ("_pad", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [("NP", C.c_int32), ("N", C.c_int32), ("P", C.c_int32), ("p_begin", C.c_int32),
                ("M", C.c_int32), ("m_out_off", C.c_int32), ("out_L", C.c_int32), ("out_pos_off", C.c_int32),
                ("ldw_m", C.c_int32), ("ldw_c", C.c_int32), ("w_m_off", C.c_int32), ("nseg", C.c_int32),
                ("epi", C.c_int32), ("M_stat", C.c_int32),
                ("out", _fp), ("W", _fp), ("bias", _fp), ("zmask", _fp),
                ("e0", _fp), ("e1", _fp), ("e2", _fp), ("partials", _fp),
                ("seg", Seg * MAX_SEG)]


class WgradArgs(C.Structure):
    _fields_ = [("NP", C.c_int32), ("N", C.c_int32), ("P", C.c_int32), ("p_begin", C.c_int32),
                ("M", C.c_int32), ("a_L", C.c_int32), ("a_pos_off", C.c_int32), ("a_m_off", C.c_int32),
                ("a_mode", C.c_int32), ("ldw_m", C.c_int32), ("ldw_c", C.c_int32), ("w_m_off", C.c_int32),
                ("nseg", C.c_int32), ("w_numel", C.c_int32), ("_pad", C.c_int32),
                ("a0", _fp), ("a1", _fp), ("ac0", _fp), ("ac1", _fp), ("ac2", _fp),
                ("w_partials", _fp), ("b_partials", _fp),
                ("b_stride", C.c_int32), ("b_off", C.c_int32),
                ("seg", Seg * MAX_SEG)]


DG_STORE, DG_MASK, DG_STATS, DG_ACCUM, DG_PREZERO = 1, 2, 4, 8, 16


class DgradOut(C.Structure):
    _fields_ = [("out", _fp), ("zmask", _fp), ("e2", _fp), ("partials", _fp), ("flags", C.c_int32), ("_pad", C.c_int32)]


class PwBwdArgs(C.Structure):
    _fields_ = [("w", WgradArgs), ("W", _fp), ("dg", DgradOut * MAX_SEG)]


class ConvtBwdArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("NP", "N", "Lin", "Lout", "K", "S", "pad", "Ci", "Co", "w_numel", "b_stride",
                                         "b_off")] + \
               [(n, _fp) for n in ("dy", "z", "ca", "cb", "cc", "src", "s_scale", "s_shift", "s_mean", "W", "dsrc", "partials",
                                   "w_partials", "b_partials")]


class BSeg(C.Structure):
    """trunet_bseg: one K-segment of a bf16 (octet layout) implicit GEMM / weight gradient"""
    _fields_ = [("src0", _fp), ("src1", _fp), ("c0", _fp), ("c1", _fp), ("c2", _fp),
                ("nchan", C.c_int32), ("L", C.c_int32), ("pos_mul", C.c_int32), ("pos_off", C.c_int32),
                ("pos_div", C.c_int32), ("mode", C.c_int32), ("kstep0", C.c_int32), ("woff", C.c_int32)]


class BGemmArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("NP", "N", "P", "p_begin", "M", "out_L", "out_pos_off", "nseg", "epi", "M_stat",
                                         "nks_total", "m_out_off")] + \
               [(n, _fp) for n in ("out", "wfrag", "bias", "zmask", "e0", "e1", "e2", "partials")] + \
               [("seg", BSeg * MAX_SEG)]


class BWgradArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("NP", "N", "P", "p_begin", "M", "a_L", "a_pos_off", "a_mode", "ldw_m", "ldw_c",
                                         "w_m_off", "nseg", "w_numel", "b_stride", "b_off", "_pad")] + \
               [(n, _fp) for n in ("a0", "a1", "ac0", "ac1", "ac2", "w_partials", "b_partials")] + \
               [("seg", BSeg * MAX_SEG)]


class BPackDesc(C.Structure):
    _fields_ = [("W", _fp), ("out", _fp)] + [(n, C.c_int32) for n in ("M", "ldw_m", "ldw_c", "w_m_off", "nseg", "nks_total")] + \
               [("nchan", C.c_int32 * MAX_SEG), ("woff", C.c_int32 * MAX_SEG), ("ks0", C.c_int32 * MAX_SEG), ("_pad", C.c_int32)]


class BDgradArgs(C.Structure):
    _fields_ = [("wfragT", _fp), ("out", _fp * MAX_SEG), ("mean", _fp * MAX_SEG), ("partials", _fp * MAX_SEG),
                ("flags", C.c_int32 * MAX_SEG), ("nrt_total", C.c_int32)]


class BPwBwdArgs(C.Structure):
    _fields_ = [("w", BWgradArgs), ("dg", BDgradArgs)]


class BConvtArgs(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("NP", "N", "Lin", "Lout", "K", "S", "pad", "Ci", "Co", "w_numel", "b_stride",
                                         "b_off", "prezero", "_pad")] + \
               [(n, _fp) for n in ("dy", "z", "ca", "cb", "cc", "src", "s_scale", "s_shift", "s_mean", "wfragT", "dsrc",
                                   "partials", "w_partials", "b_partials")]


MAX_RES = 8


class LossArgs(C.Structure):
    """trunet_loss_args"""
    _fields_ = [("l1_partials", _fp), ("parts", _fp * MAX_RES), ("n_l1", C.c_int32), ("nres", C.c_int32),
                ("nrows", C.c_int32 * MAX_RES), ("l1_count", C.c_double), ("count", C.c_double * MAX_RES),
                ("sc_lambda", C.c_float), ("mag_lambda", C.c_float), ("stft_lambda", C.c_float), ("_pad", C.c_float)]


class LossGatherArgs(C.Structure):
    """trunet_loss_gather_args"""
    _fields_ = [("fr_sc", _fp * MAX_RES), ("fr_mag", _fp * MAX_RES), ("n", C.c_int32 * MAX_RES), ("hop", C.c_int32 * MAX_RES),
                ("win_length", C.c_int32 * MAX_RES), ("nres", C.c_int32), ("_pad", C.c_int32)]


_lib = None


class TrunetHipError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TrunetHipError(
                "libtrunet_hip.so not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`; "
                "there is no CPU fallback for the HIP hot path." % LIB_PATH)
        _lib = C.CDLL(LIB_PATH)
        _declare(_lib)
    return _lib


def _declare(L):
    i, d, f, p, i64 = C.c_int, C.c_double, C.c_float, _fp, C.c_int64
    sig = {
        "trunet_conv_gemm_nparts": [i],
        "trunet_gemm_x3_enable": [i],
        "trunet_conv_gemm": [C.POINTER(GemmArgs), p],
        "trunet_conv_gemm_plan": [C.POINTER(GemmArgs)] + [C.POINTER(C.c_int)] * 6,
        "trunet_conv_wgrad_nparts": [],
        "trunet_conv_wgrad": [C.POINTER(WgradArgs), p],
        "trunet_convt_bwd_nparts": [],
        "trunet_convt_bwd": [C.POINTER(ConvtBwdArgs), p],
        "trunet_pw_bwd_nparts": [],
        "trunet_pw_bwd": [C.POINTER(PwBwdArgs), p],
        "trunet_reduce_partials": [p, p, i, i, i, p],
        "trunet_bn_finalize_fwd": [p, i, i, d, p, p, f, f, p, p, p, p, p, p, p, p],
        "trunet_relu_bwd_stats_nparts": [],
        "trunet_relu_bwd_stats": [p, p, p, p, p, p, i, i, i, i, p],
        "trunet_bn_eval_affine": [i, p, p, p, p, f, p, p, p],
        "trunet_bn_finalize_bwd": [p, i, i, d, p, p, p, p, p, p, p, p, p],
        "trunet_to_frames_last": [p, p, i, i, i, i, p],
        "trunet_from_frames_last": [p, p, i, i, i, i, p],
        "trunet_from_frames_last_affine": [p, p, i, i, i, i, p, p, i, p],
        "trunet_conv_first_fwd": [p, p, p, p, i, i, i, i, i, i, i, p],
        "trunet_dwconv_nparts": [i],
        "trunet_dwconv_bwd_nparts": [i],
        "trunet_dwconv_fwd": [p, p, p, p, p, p, p, i, i, i, i, i, i, i, p],
        "trunet_dwconv_bwd": [p] * 14 + [i] * 7 + [p],
        "trunet_dwconv_bwd_rz": [p] * 14 + [i] * 7 + [p],
        "trunet_gru_fwd": [p, p, p, p, p, p, p, i, i, i, p],
        "trunet_gru_bwd": [p, p, p, p, p, p, p, i, i, i, i, p],
        "trunet_gru_cell": [p, p, p, p, i, i, i, p],
        "trunet_to_seq_major": [p, p, p, p, i, i, i, i, i, i, i, p],
        "trunet_from_seq_major_nparts": [i, i, i],
        "trunet_from_seq_major": [p, p, p, p, p, p, p, i, i, i, i, i, i, p],
        "trunet_tgru_cell_fwd": [p, p, p, p, i, i, i, i, p],
        "trunet_tgru_rec_fwd": [p, p, p, p, p, i, i, i, p],
        "trunet_tgru_rec_bwd": [p, p, p, p, p, p, i, i, i, i, p],
        "trunet_tgru_cell_bwd": [p, p, p, p, p, p, i, i, i, i, i, p],
        "trunet_adamw": [p, p, p, p, i64, f, f, f, f, f, i, p],
        "trunet_sumsq": [p, i64, p, p],
        "trunet_checksum_batch": [p, i, p, p],
        "trunet_stft_features": [p, p, p, p, i, i, i, i, p],
        "trunet_pcen": [p, p, i, i, i, f, f, f, f, f, p],
        "trunet_mask_istft_fwd": [p, p, p, p, p, p, i, i, i, f, p],
        "trunet_mask_istft_l1_nparts": [i, i],
        "trunet_mask_istft_bwd": [p, p, p, p, i, i, i, f, p],
        "trunet_l1_grad": [p, p, p, p, i64, p],
        "trunet_reduce_cols": [p, i, i, p, p],
        "trunet_stft_loss_fwd": [p, p, p, p, p, i, i, i, i, p],
        "trunet_stft_mag": [p, p, p, p, p, p, i, i, i, i, p],
        "trunet_stft_loss_fwdgrad": [p, p, p, p, p, p, p, i, i, i, i, i, p],
        "trunet_stream_features": [p, p, p, p, p, i, i, i, f, f, f, f, f, p],
        "trunet_stream_mask_istft": [p, p, p, p, i, f, f, p],
        "trunet_loss_scratch_bytes": [],
        "trunet_loss_finalize": [C.POINTER(LossArgs), p, p, p, p],
        "trunet_loss_grad_gather": [C.POINTER(LossGatherArgs), p, p, p, p, p, i, i, p],
        "trunet_stft_loss_bwd_gather": [p, p, p, p, p, p, p, i, i, i, i, i, p],
        "trunet_stft_mag_bwd": [p, p, p, p, p, p, i, i, i, i, i, p],
        "trunet_phm_fwd": [p, p, p, i64, f, p],
        "trunet_phm_bwd": [p, p, p, p, p, i64, f, p],
        "trunet_augment_mix": [p, p, p, p, p, i, i, p],
        "trunet_stream_fwd_grid": [i],
        "trunet_stream_fwd_scratch_floats": [i],
        "trunet_stream_fwd_check": [C.POINTER(C.c_int32), i, i64, i],
        "trunet_stream_fwd": [p, p, p, C.POINTER(C.c_int32), i, i64, p, p, p, i, i, p],
        "trunet_stream_fwd_x3_mask": [],
        "trunet_stream_fwd_x3_check": [C.POINTER(C.c_int32), i, i64, i],
        "trunet_stream_fwd_x3": [p, p, p, C.POINTER(C.c_int32), i, i64, p, p, p, i, i, p],
        "trunet_bf16_gemm_nparts": [],
        "trunet_bf16_gemm": [C.POINTER(BGemmArgs), p],
        "trunet_bf16_pack_weight": [p, p, i, i, i, i, i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), p],
        "trunet_bf16_pack_weights_batch": [p, i, i, p],
        "trunet_bf16_wgrad": [C.POINTER(BWgradArgs), p],
        "trunet_bf16_pw_bwd_nparts": [],
        "trunet_bf16_pw_bwd": [C.POINTER(BPwBwdArgs), p],
        "trunet_bf16_convt_bwd_nparts": [],
        "trunet_bf16_convt_bwd": [C.POINTER(BConvtArgs), p],
        "trunet_bf16_dw_nparts": [i, i],
        "trunet_bf16_dwconv_fwd": [p, p, p, p, p, p, p, i, i, i, i, i, i, i, p],
        "trunet_bf16_dwconv_bwd": [p] * 14 + [i] * 7 + [p],
        "trunet_bf16_gru_fwd": [p, p, p, p, p, p, p, i, i, i, p],
        "trunet_bf16_gru_bwd": [p, p, p, p, p, p, p, i, i, i, p],
        "trunet_bf16_from_frames_last": [p, p, i, i, i, p],
        "trunet_bf16_from_ncl": [p, p, i, i, i, i, p],
        "trunet_bf16_to_ncl": [p, p, i, i, i, i, p],
        "trunet_bf16_to_frames_last": [p, p, i, i, i, p],
        "trunet_debug_mfma_peak": [p, i, i, p],
    }
    for name, args in sig.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = C.c_int
    L.trunet_stream_fwd_scratch_floats.restype = C.c_size_t
    L.trunet_loss_scratch_bytes.restype = C.c_size_t
    L._declared = sorted(sig)


def declared_symbols():
    """Names of every entry point this binding expects (checked against the header in tests)."""
    return list(lib()._declared)


X3_GEMM, X3_BWD = 1, 2                   # trunet_hip.h: TRUNET_X3_GEMM, TRUNET_X3_BWD
_MFMA_KINDS = {"fp32": 0, "bf16x3-bwd": X3_BWD, "bf16x3-fwd": X3_GEMM, "bf16x3": X3_GEMM | X3_BWD}


def set_fp32_mfma(kind):
    """Which matrix instruction the fp32 GEMM kernels multiply on (storage, accumulation and results are fp32 in every case).
    "fp32": v_mfma_f32_32x32x2_f32 everywhere.  "bf16x3-bwd" (the default since round 4): the fused backward kernels
    (pw_bwd.hip, convt_bwd_x3.hip) split their fp32 operands into three bf16 terms and multiply on the bf16 MFMA -- backward
    is linear in the saved forward state, so the forward pass, the loss and the loss gradient are bit for bit those of
    "fp32" and the parameter gradients move at the 1e-7 level.  "bf16x3" (opt-in): the forward implicit GEMMs as well
    (gemm_x3.hip): same accuracy against float64, different rounding pattern of the network output.
    Process-wide; returns the previous kind."""
    if kind not in _MFMA_KINDS:
        raise ValueError("fp32 MFMA kind must be one of %s, got %r" % (sorted(_MFMA_KINDS), kind))
    prev = lib().trunet_gemm_x3_enable(_MFMA_KINDS[kind])
    return [k for k, v in _MFMA_KINDS.items() if v == prev][0]


def fp32_mfma():
    m = lib().trunet_gemm_x3_enable(-1)
    return [k for k, v in _MFMA_KINDS.items() if v == m][0]


def check(rc, what=""):
    if rc != 0:
        raise TrunetHipError("libtrunet_hip call failed (%s): code %d" % (what, rc))


def ptr(t):
    """Device pointer of a contiguous fp32 torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


def ptr16(t):
    """Device pointer of a contiguous bf16 torch tensor in the octet layout (or None)."""
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous(), (t.device, t.dtype, t.is_contiguous())
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def make_seg(src0, nchan, L, pos_mul=1, pos_off=0, pos_div=1, woff=0, mode=PRO_NONE, src1=None, c0=None, c1=None,
             c2=None):
    s = Seg()
    s.src0, s.src1 = ptr(src0), ptr(src1)
    s.c0, s.c1, s.c2 = ptr(c0), ptr(c1), ptr(c2)
    s.nchan, s.L = nchan, L
    s.pos_mul, s.pos_off, s.pos_div = pos_mul, pos_off, pos_div
    s.woff, s.mode = woff, mode
    return s


# ---- flat gradient registry: the TRU-Net backward leaves every parameter gradient as a view of ONE flat tensor (gaps
# between parameters are zero); the gradient all-reduce and FusedAdamW look the buffer up here by storage address and
# work on it in place instead of packing the ~100 gradients again.
_FLAT = {}          # storage address -> (flat tensor, layout, total); the newest few are kept alive (1.2 MB each)


# Weights and BatchNorm buffers are also written through raw device pointers (FusedAdamW, bn_finalize_fwd), which does
# not move torch's tensor version counters: everything that caches a function of them (TRUNet.folded) keys on this
# epoch as well, and every such writer bumps it.
_MUTATION_EPOCH = [0]


def mutation_epoch():
    return _MUTATION_EPOCH[0]


def bump_mutation_epoch():
    _MUTATION_EPOCH[0] += 1


def register_flat_grad(flat, layout, total):
    """flat: 1-D fp32 tensor of ``total`` elements; layout: {id(parameter): element offset}"""
    while len(_FLAT) >= 4:
        del _FLAT[next(iter(_FLAT))]
    _FLAT[flat.untyped_storage().data_ptr()] = (flat, layout, total)


def flat_grad_of(grad):
    """(flat tensor, layout, total) if ``grad`` is a view of a registered flat gradient"""
    return _FLAT.get(grad.untyped_storage().data_ptr())


_TW = {}


def twiddles(n, device):
    """exp(-2*pi*i*t/n), t < n/2, as an (n/2, 2) fp32 device tensor (computed once in fp64)."""
    key = (n, str(device))
    if key not in _TW:
        import numpy as np
        t = np.arange(n // 2, dtype=np.float64)
        w = np.stack([np.cos(2 * np.pi * t / n), -np.sin(2 * np.pi * t / n)], 1)
        _TW[key] = torch.tensor(w, dtype=torch.float32, device=device).contiguous()
    return _TW[key]
