"""ctypes binding of libaq_hip.so (C ABI: include/aq_hip.h).

There is no fallback: if the library is missing, loading raises; if a call fails, the message of
aq_last_error() is raised as RuntimeError.  Tensors handed to the kernels must be CUDA(HIP), fp32
(uint8 for codes) and contiguous; anything else raises ValueError before the call.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AQ_HIP_LIB", os.path.join(_HERE, "lib", "libaq_hip.so"))  # override: development builds
ABI_VERSION = 13
FLAG_EXACT = 1
BN_RELU, BN_EVAL = 1, 2
MAX_LEVELS = 128
STATIC_SCRATCH_BYTES = 2 << 20

EXPORTS = (
    "aq_abi_version", "aq_last_error", "aq_last_kernel", "aq_synth_fwd", "aq_synth_bwd", "aq_synth_workspace_bytes",
    "aq_conv_fwd", "aq_conv_fwd_codes", "aq_conv_dgrad", "aq_conv_wgrad", "aq_conv_wgrad_workspace_bytes",
    "aq_linear_fwd", "aq_linear_bwd", "aq_conv_fwd_add", "aq_bn_relu_fwd", "aq_bn_relu_bwd", "aq_bn_workspace_bytes",
    "aq_shortcut_fwd", "aq_shortcut_bwd", "aq_conv_wgrad_partial", "aq_conv_wgrad_partial_group", "aq_conv_wgrad_reduce", "aq_synth_bwd_adam",
    "aq_packed_bytes", "aq_pack_codes", "aq_unpack_codes", "aq_augment_batch", "aq_absmax", "aq_bn_relu_pool_fwd", "aq_bn_relu_pool_bwd",
    "aq_bnconv_supported", "aq_bnconv_fwd", "aq_bnconv_wgrad_partial", "aq_sumsq", "aq_sumsq_workspace_bytes", "aq_ce_loss", "aq_ce_loss_workspace_bytes",
    "aq_conv_stats_bytes", "aq_conv_fwd_stats", "aq_conv_dgrad_bn_bytes", "aq_conv_dgrad_bn",
    "aq_conv_bwd_sums_bytes", "aq_conv_bwd", "aq_adam_step",
    "aq_bn_relu_bwd_shortcut", "aq_bn_relu_fwd_shortcut", "aq_conv_dgrad_add", "aq_synth_fwd_static",
    "aq_tern_workspace_bytes", "aq_tern_fwd", "aq_tern_bwd", "aq_inq_partition", "aq_tern_bwd_adam", "aq_inq_effective", "aq_inq_effective_sumsq", "aq_bn_add_relu_fwd", "aq_bn_add_relu_bwd",
    "aq_pkt_workspace_bytes", "aq_pkt_loss", "aq_pkt_kernel_matrix", "aq_maxpool3s2_fwd", "aq_maxpool3s2_bwd",
)
TERN_TWN, TERN_TTQ, TERN_INQ = 0, 1, 2


class AqSeg(C.Structure):
    _fields_ = [
        ("p_c", C.c_void_p), ("q", C.c_void_p), ("rows", C.c_int64), ("L", C.c_int32), ("reserved", C.c_int32),
        ("W", C.c_void_p), ("idx", C.c_void_p), ("W_hard", C.c_void_p), ("sumsq", C.c_void_p),
        ("dW", C.c_void_p), ("dsumsq", C.c_void_p), ("dp_c", C.c_void_p), ("dq", C.c_void_p),
        ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_step", C.c_void_p),
    ]


class AqTernSeg(C.Structure):
    _fields_ = [("W", C.c_void_p), ("n", C.c_int64), ("codes", C.c_void_p), ("packed", C.c_void_p), ("Wt", C.c_void_p),
                ("stats", C.c_void_p), ("q", C.c_void_p), ("Wp", C.c_void_p), ("Wn", C.c_void_p), ("mask", C.c_void_p),
                ("dWt", C.c_void_p), ("dW", C.c_void_p), ("dWp", C.c_void_p), ("dWn", C.c_void_p),
                ("frozen", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_step", C.c_void_p),
                ("Wp_m", C.c_void_p), ("Wp_v", C.c_void_p), ("Wp_step", C.c_void_p),
                ("Wn_m", C.c_void_p), ("Wn_v", C.c_void_p), ("Wn_step", C.c_void_p)]


class AqConvStats(C.Structure):
    _fields_ = [("P", C.c_int32), ("ntiles", C.c_int32), ("tile_px", C.c_int32)]


class AqAdamItem(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("step", C.c_void_p),
                ("n", C.c_int64), ("reg", C.c_float), ("reserved", C.c_int32)]


class AqWgradJob(C.Structure):
    _fields_ = [("part", C.c_void_p), ("dW", C.c_void_p), ("n", C.c_int32), ("nslab", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C quantized-cnn_amd/csrc`); there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i, f, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
        L.aq_abi_version.restype = i
        L.aq_last_error.restype = C.c_char_p
        L.aq_synth_fwd.argtypes = [C.POINTER(AqSeg), i, f, C.c_uint, vp, sz, vp]
        L.aq_synth_fwd_static.argtypes = [C.POINTER(AqSeg), i, f, C.c_uint, vp]
        L.aq_synth_fwd_static.restype = i
        L.aq_synth_bwd.argtypes = [C.POINTER(AqSeg), i, f, f, C.c_uint, vp, sz, vp]
        L.aq_synth_bwd_adam.argtypes = [C.POINTER(AqSeg), i, f, f, C.c_double, C.c_double, C.c_double, C.c_double, C.c_uint, vp, sz, vp]
        L.aq_synth_bwd_adam.restype = i
        L.aq_synth_workspace_bytes.argtypes = [C.POINTER(AqSeg), i, i]
        L.aq_synth_workspace_bytes.restype = sz
        conv_dims = [i] * 8
        L.aq_conv_fwd.argtypes = [vp, vp, vp] + conv_dims + [vp]
        L.aq_conv_fwd_add.argtypes = [vp, vp, vp, vp] + conv_dims + [vp]
        L.aq_conv_fwd_codes.argtypes = [vp, vp, vp, i, vp] + conv_dims + [vp]
        L.aq_conv_dgrad.argtypes = [vp, vp, vp] + conv_dims + [vp]
        L.aq_conv_dgrad_add.argtypes = [vp, vp, vp, vp] + conv_dims + [vp]
        L.aq_conv_dgrad_add.restype = i
        L.aq_conv_wgrad.argtypes = [vp, vp, vp] + conv_dims + [vp, sz, vp]
        L.aq_conv_wgrad_partial.argtypes = [vp, vp] + conv_dims + [vp, sz, C.POINTER(C.c_int), vp]
        L.aq_conv_wgrad_partial_group.argtypes = [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.c_int] + conv_dims + [sz, C.POINTER(C.c_int), vp]
        L.aq_conv_wgrad_reduce.argtypes = [C.POINTER(AqWgradJob), i, vp]
        L.aq_conv_wgrad_workspace_bytes.argtypes = conv_dims
        L.aq_conv_wgrad_workspace_bytes.restype = sz
        L.aq_linear_fwd.argtypes = [vp, vp, vp, vp, i, i, i, vp]
        L.aq_linear_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, vp]
        L.aq_bn_relu_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, f, f, C.c_uint, vp, vp, vp, vp, sz, C.POINTER(AqConvStats), vp]
        L.aq_conv_stats_bytes.argtypes = conv_dims
        L.aq_conv_stats_bytes.restype = sz
        L.aq_conv_fwd_stats.argtypes = [vp, vp, vp, vp] + conv_dims + [vp, sz, C.POINTER(AqConvStats), vp]
        L.aq_conv_fwd_stats.restype = i
        L.aq_bn_relu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, C.c_uint, vp, vp, vp, vp, sz, C.POINTER(AqConvStats), vp]
        L.aq_conv_dgrad_bn_bytes.argtypes = conv_dims
        L.aq_conv_dgrad_bn_bytes.restype = sz
        L.aq_conv_dgrad_bn.argtypes = [vp, vp, vp] + conv_dims + [vp, vp, vp, vp, vp, vp, sz, C.POINTER(AqConvStats), vp]
        L.aq_conv_dgrad_bn.restype = i
        L.aq_conv_bwd_sums_bytes.argtypes = conv_dims
        L.aq_conv_bwd_sums_bytes.restype = sz
        L.aq_conv_bwd.argtypes = [vp, vp, vp, vp] + conv_dims + [vp, vp, vp, vp, vp, vp, sz, C.POINTER(AqConvStats), vp, sz, C.POINTER(C.c_int), vp]
        L.aq_conv_bwd.restype = i
        L.aq_adam_step.argtypes = [C.POINTER(AqAdamItem), i, C.c_double, C.c_double, C.c_double, C.c_double, vp]
        L.aq_adam_step.restype = i
        L.aq_bn_relu_bwd_shortcut.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, i, i, i, C.c_uint, vp, vp, vp, vp, sz, C.POINTER(AqConvStats), vp]
        L.aq_bn_relu_bwd_shortcut.restype = i
        L.aq_bn_relu_fwd_shortcut.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, i, f, f, C.c_uint, vp, vp, vp, vp, sz, C.POINTER(AqConvStats),
                                              i, i, vp, vp]
        L.aq_bn_relu_fwd_shortcut.restype = i
        L.aq_shortcut_fwd.argtypes = [vp, vp, i, i, i, i, i, i, vp]
        L.aq_shortcut_bwd.argtypes = [vp, vp, i, i, i, i, i, i, vp]
        L.aq_maxpool3s2_fwd.argtypes = [vp, vp, vp, i, i, i, i, vp]
        L.aq_maxpool3s2_bwd.argtypes = [vp, vp, vp, i, i, i, i, vp]
        L.aq_packed_bytes.argtypes = [C.c_longlong, i]
        L.aq_packed_bytes.restype = sz
        L.aq_pack_codes.argtypes = [vp, C.c_longlong, i, vp, vp]
        L.aq_unpack_codes.argtypes = [vp, C.c_longlong, i, vp, vp]
        L.aq_pack_codes.restype = L.aq_unpack_codes.restype = i
        L.aq_augment_batch.argtypes = [vp, C.c_longlong, vp, vp, vp, vp, i, i, i, i, i, C.POINTER(f), C.POINTER(f), vp, vp]
        L.aq_absmax.argtypes = [C.POINTER(vp), C.POINTER(C.c_longlong), i, vp, vp]
        L.aq_augment_batch.restype = L.aq_absmax.restype = i
        L.aq_bn_relu_pool_fwd.argtypes = L.aq_bn_relu_fwd.argtypes
        L.aq_bn_relu_pool_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, C.c_uint, vp, vp, vp, vp, sz, vp]
        L.aq_bn_relu_pool_fwd.restype = L.aq_bn_relu_pool_bwd.restype = i
        L.aq_bnconv_supported.argtypes = conv_dims
        L.aq_bnconv_fwd.argtypes = [vp, vp, vp, vp, vp, vp, f, f, vp, vp, vp, vp, vp] + conv_dims + [vp, sz, vp]
        L.aq_bnconv_wgrad_partial.argtypes = [vp, vp, vp, vp, vp, vp] + conv_dims + [vp, sz, C.POINTER(C.c_int), vp]
        L.aq_bnconv_supported.restype = L.aq_bnconv_fwd.restype = L.aq_bnconv_wgrad_partial.restype = i
        L.aq_sumsq.argtypes = [C.POINTER(vp), C.POINTER(C.c_longlong), i, vp, vp, sz, vp]
        L.aq_sumsq_workspace_bytes.argtypes = [C.POINTER(C.c_longlong), i]
        L.aq_sumsq_workspace_bytes.restype = sz
        L.aq_ce_loss.argtypes = [vp, vp, i, i, vp, i, vp, i, f, f, vp, vp, vp, sz, vp]
        L.aq_ce_loss_workspace_bytes.argtypes = [i, i]
        L.aq_ce_loss_workspace_bytes.restype = sz
        L.aq_sumsq.restype = L.aq_ce_loss.restype = i
        L.aq_tern_workspace_bytes.argtypes = [C.POINTER(AqTernSeg), i]
        L.aq_tern_workspace_bytes.restype = sz
        L.aq_tern_fwd.argtypes = [C.POINTER(AqTernSeg), i, i, f, i, vp, sz, vp]
        L.aq_tern_bwd.argtypes = [C.POINTER(AqTernSeg), i, i, vp, sz, vp]
        L.aq_inq_partition.argtypes = [vp, vp, vp, C.c_longlong, f, i, i, vp]
        L.aq_tern_bwd_adam.argtypes = [C.POINTER(AqTernSeg), i, i, C.c_double, C.c_double, C.c_double, C.c_double, f, vp, sz, vp]
        L.aq_inq_effective.argtypes = [C.POINTER(AqTernSeg), i, vp]
        L.aq_inq_effective_sumsq.argtypes = [C.POINTER(AqTernSeg), i, vp, vp, sz, vp]
        L.aq_inq_effective_sumsq.restype = i
        L.aq_tern_fwd.restype = L.aq_tern_bwd.restype = L.aq_inq_partition.restype = L.aq_tern_bwd_adam.restype = L.aq_inq_effective.restype = i
        L.aq_pkt_workspace_bytes.argtypes = [i, i, i]
        L.aq_pkt_workspace_bytes.restype = sz
        L.aq_pkt_loss.argtypes = [vp, vp, i, i, i, f, vp, vp, vp, vp, sz, vp]
        L.aq_pkt_kernel_matrix.argtypes = [vp, i, i, f, vp, vp, sz, vp]
        L.aq_pkt_loss.restype = L.aq_pkt_kernel_matrix.restype = i
        L.aq_bn_workspace_bytes.argtypes = [i, i, i]
        L.aq_bn_workspace_bytes.restype = sz
        for name in ("aq_synth_fwd", "aq_synth_bwd", "aq_conv_fwd", "aq_conv_fwd_add", "aq_conv_fwd_codes", "aq_conv_dgrad",
                     "aq_conv_wgrad", "aq_linear_fwd", "aq_linear_bwd", "aq_bn_relu_fwd", "aq_bn_relu_bwd", "aq_shortcut_fwd", "aq_shortcut_bwd", "aq_conv_wgrad_partial", "aq_conv_wgrad_reduce"):
            getattr(L, name).restype = i
        L.aq_last_kernel.restype = C.c_char_p
        L.aq_bn_add_relu_fwd.argtypes = [vp, vp, vp, vp, vp, vp, i, i, i, f, f, vp, vp, vp, vp, vp, sz, vp, vp]
        L.aq_bn_add_relu_bwd.argtypes = [vp, vp, vp, vp, vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, sz, vp]
        L.aq_bn_add_relu_fwd.restype = L.aq_bn_add_relu_bwd.restype = i
        if L.aq_abi_version() != ABI_VERSION:
            raise ImportError("libaq_hip.so ABI %d != binding ABI %d: rebuild" % (L.aq_abi_version(), ABI_VERSION))
        _lib = L
    return _lib


def last_kernel():
    """Name, with template arguments, of the main kernel the most recent synthesis / ternary call of this thread launched."""
    return lib().aq_last_kernel().decode()


def check(rc):
    if rc != 0:
        raise RuntimeError("libaq_hip: %s (code %d)" % (lib().aq_last_error().decode(), rc))


def ptr(t, dtype=torch.float32, name="tensor"):
    """Device pointer of a tensor after checking it is what the kernels assume."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor):
        raise ValueError("%s: expected a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s lives on %s: the quantized layers run on the HIP device only "
                           "(no CPU path); move the model and its inputs to cuda" % (name, t.device))
    if t.dtype != dtype:
        raise ValueError("%s: dtype %s, expected %s" % (name, t.dtype, dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
