"""ctypes binding of libasr_hip.so -- the only door from Python into the HIP kernels.

The library is built in-tree by ``make -C chainer-speech-recognition_amd`` (or ``__graft_entry__.build()``).
There is NO fallback: if the shared object is missing or a symbol is absent, importing an op raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# ASR_HIP_LIB: another build of the same library (what-if builds of the timing tools); ASR_ACT=f16 (or float16 / half): the in-tree
# IEEE-half build libasr_hip_f16.so (BASELINE configs[4]'s "fp16 MFMA": csrc/common.hpp); the default is the in-tree bfloat16 build.
# Read once, when the first asr module is imported: one process, one activation format.
_ACT = os.environ.get("ASR_ACT", "").lower()
LIB_PATH = os.environ.get("ASR_HIP_LIB") or os.path.join(os.path.dirname(_HERE),
                                                        "libasr_hip_f16.so" if _ACT in ("f16", "fp16", "float16", "half") else "libasr_hip.so")

_lib = None


def act_dtype():
    """torch dtype of the 16-bit activation format of the loaded library (asr_act_dtype): bfloat16, or float16 for libasr_hip_f16.so.
    The modules bind it once at import under the historical name BF16."""
    return torch.float16 if lib().asr_act_dtype() else torch.bfloat16


def debug_flag(key, default):
    """integer value of `key` in ASR_DEBUG="key=value,key=value" -- the ONE table of what-if switches (csrc/common.hpp: debug_flag lists
    the keys the library reads; this layer reads tn_group, gru_gates_f16, side_join, side_priority, conv_mp).  Nothing in a normal run sets it."""
    for item in os.environ.get("ASR_DEBUG", "").split(","):
        k, _, v = item.strip().partition("=")
        if k == key and v:
            return int(v)
    return default

c_void_p, c_int, c_float, c_size_t = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
c_longlong = ctypes.c_longlong

# name -> (restype, argtypes).  Mirrors include/asr_hip.h one to one; tests/test_abi.py checks both ways.
SIGNATURES = {
    "asr_version": (c_int, []),
    "asr_act_dtype": (c_int, []),
    "asr_stream_delay": (c_int, [c_void_p, c_int]),
    "asr_occupy_cus": (c_int, [c_void_p, c_int, c_int, c_int]),
    "asr_stream_traffic": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_longlong]),
    "asr_ctc_workspace_bytes": (c_size_t, [c_int] * 5),
    "asr_ctc_forward": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 3 + [c_size_t]),
    "asr_ctc_forward_lse": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_void_p] * 3 + [c_size_t, c_void_p]),
    "asr_ctc_backward": (c_int, [c_void_p] * 3 + [c_int] * 5 + [c_void_p, c_int, c_float, c_void_p, c_void_p, c_size_t]),
    "asr_ctc_loss_grad": (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_float] + [c_void_p] * 4 + [c_size_t]),
    "asr_specgram": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_longlong] + [c_int] * 4 + [c_float, c_void_p, c_void_p, c_int,
                               c_void_p, c_void_p, c_int, c_void_p]),
    "asr_mel_bands_bytes": (c_size_t, []),
    "asr_mel_bands": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_size_t]),
    "asr_specgram_bands": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_longlong] + [c_int] * 4 + [c_float, c_void_p, c_void_p, c_int,
                                     c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "asr_logmel": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int, c_void_p]),
    "asr_deltas": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 3),
    "asr_batchnorm_stats": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_float, c_float] + [c_void_p] * 5),
    "asr_rsqrt_eps": (c_int, [c_void_p, c_void_p, c_float, c_void_p, c_int]),
    "asr_batchnorm_fwd": (c_int, [c_void_p] * 6 + [c_longlong, c_int, c_void_p]),
    "asr_batchnorm_bwd": (c_int, [c_void_p] * 6 + [c_longlong, c_int] + [c_void_p] * 4),
    "asr_cmn_pspec": (c_int, [c_void_p] * 3 + [c_int] * 3),
    "asr_add_white_noise": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_void_p, ctypes.c_ulonglong]),
    "asr_augment_specgram": (c_int, [c_void_p] * 5 + [c_int] * 4 + [c_void_p]),
    "asr_running_stats_update": (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_longlong] + [c_void_p] * 4),
    "asr_normalize_bcmt": (c_int, [c_void_p] * 4 + [c_int] * 3),
    "asr_argmax_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "asr_ctc_collapse": (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 2),
    "asr_edit_distance": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "asr_gemm_nt": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 4),
    "asr_gemm_nt_8ph_ok": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int]),
    "asr_gemm_nt_8ph": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int]),
    "asr_conv_nt": (c_int, [c_void_p] * 3 + [c_int, c_void_p, c_int, c_void_p] + [c_int] * 12),
    "asr_conv_nt_8ph_ok": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 9),
    "asr_conv_nt_8ph": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 12),
    "asr_conv_nt_8pn_ok": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 9),
    "asr_conv_nt_8pn": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p] + [c_int] * 12),
    "asr_conv_direct_ok": (c_int, [c_int] * 11),
    "asr_conv_direct_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 12),
    "asr_conv_mp_ok": (c_int, [c_int] * 5),
    "asr_conv_mp_fwd": (c_int, [c_void_p] * 3 + [c_int] + [c_void_p] * 3 + [c_int] * 11),
    "asr_conv_mp_bwd_workspace": (c_longlong, [c_int] * 5),
    "asr_conv_mp_bwd": (c_int, [c_void_p] * 7 + [c_int] * 12),
    "asr_pack_input_pad": (c_int, [c_void_p, c_void_p, c_int] + [c_longlong] * 4 + [c_int] * 5 + [c_void_p]),
    "asr_conv_weight_pack_bwd": (c_int, [c_void_p] * 3 + [c_int] * 4),
    "asr_gemm_tn_acc": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int] + [c_int] * 3),
    "asr_gemm_tn_acc_group": (c_int, [c_void_p, c_int] + [c_void_p] * 9),
    "asr_gemm_tn_8ph_ok": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int]),
    "asr_gemm_tn_acc_group_8ph": (c_int, [c_void_p, c_int] + [c_void_p] * 9),
    "asr_cast_bf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int]),
    "asr_cast_bf16_many": (c_int, [c_void_p, c_void_p, c_int, c_longlong]),
    "asr_bf16_to_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong]),
    "asr_permute4": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int] + [c_int] * 4 + [c_longlong] * 4),
    "asr_im2col": (c_int, [c_void_p, c_void_p, c_int] + [c_longlong] * 4 + [c_int] * 10 + [c_void_p]),
    "asr_col2im": (c_int, [c_void_p, c_void_p] + [c_int] * 10 + [c_void_p]),
    "asr_activation_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_float]),
    "asr_activation_bwd": (c_int, [c_void_p] * 4 + [c_longlong, c_int, c_float]),
    "asr_glu_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int]),
    "asr_glu_bwd": (c_int, [c_void_p] * 4 + [c_longlong, c_int]),
    "asr_dropout": (c_int, [c_void_p] * 3 + [c_longlong, c_float, ctypes.c_uint]),
    "asr_conv_weight_pack": (c_int, [c_void_p] * 3 + [c_int] * 6),
    "asr_conv_weight_grad_unpack": (c_int, [c_void_p] * 3 + [c_int] * 6),
    "asr_conv_tn_acc": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 12),
    "asr_conv_tn_8ph_ok": (c_int, [c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 10),
    "asr_conv_tn_acc_8ph": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 12),
    "asr_conv_tn_copies": (c_int, [c_int] * 4),
    "asr_conv_tn_acc_copies": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p] + [c_int] * 13),
    "asr_conv_weight_grad_unpack_copies": (c_int, [c_void_p, c_void_p, c_int, c_void_p] + [c_int] * 6),
    "asr_maxout2_fwd": (c_int, [c_void_p] * 3 + [c_longlong]),
    "asr_maxout2_bwd": (c_int, [c_void_p] * 4 + [c_longlong]),
    "asr_maxout2_pool_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int, c_int]),
    "asr_maxout2_pool_bwd": (c_int, [c_void_p] * 4 + [c_longlong, c_int, c_int, c_int]),
    "asr_maxout2_pool_bwd_db": (c_int, [c_void_p] * 5 + [c_longlong, c_int, c_int, c_int]),
    "asr_maxout2_pool_bwd_db_ok": (c_int, [c_int]),
    "asr_maxpool_h_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int, c_int]),
    "asr_maxpool_h_bwd": (c_int, [c_void_p] * 4 + [c_longlong, c_int, c_int, c_int]),
    "asr_add_bf16": (c_int, [c_void_p] * 4 + [c_longlong]),
    "asr_colsum_acc": (c_int, [c_void_p, c_void_p, c_int, c_longlong, c_int, c_int, c_void_p]),
    "asr_layernorm_fwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 4 + [c_longlong, c_int, c_int]),
    "asr_layernorm_fwd_lse": (c_int, [c_void_p] * 8 + [c_longlong, c_int]),
    "asr_layernorm_fwd_lse_ok": (c_int, [c_int, c_int]),
    "asr_layernorm_bwd": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int] + [c_void_p] * 3 + [c_void_p, c_int, c_void_p, c_void_p,
                                  c_longlong, c_int, c_int]),
    "asr_layernorm_bwd_rows_ws_bytes": (c_longlong, [c_longlong, c_int]),
    "asr_layernorm_bwd_rows": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_void_p, c_longlong, c_int, c_int, c_void_p, c_longlong]),
    "asr_layernorm_fold_partials": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "asr_layernorm_ctc_bwd_ws_bytes": (c_longlong, [c_int, c_int, c_int]),
    "asr_layernorm_ctc_bwd": (c_int, [c_void_p] * 7 + [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_longlong, c_int] +
                              [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_float] * 2 + [c_void_p]),
    "asr_weightnorm_fwd": (c_int, [c_void_p] * 5 + [c_int, c_int]),
    "asr_weightnorm_bwd": (c_int, [c_void_p] * 7 + [c_int, c_int]),
    "asr_channel_stats": (c_int, [c_void_p, c_void_p, c_longlong, c_int, c_void_p, c_void_p]),
    "asr_channel_affine": (c_int, [c_void_p] * 5 + [c_longlong, c_int]),
    "asr_weightnorm_init": (c_int, [c_void_p] * 5 + [c_int]),
    "asr_gru_sync_bytes": (c_size_t, [c_int] * 3),
    "asr_gru_fwd_accepts_bf16_gi": (c_int, [c_int] * 5),
    "asr_gru_gates_f16_ok": (c_int, [c_int] * 5),
    "asr_gru_fwd": (c_int, [c_void_p, c_void_p, c_int] + [c_void_p] * 6 + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_int]),
    "asr_gru_bwd": (c_int, [c_void_p] * 10 + [c_int] * 4 + [c_void_p, c_int, c_void_p, c_void_p, c_int]),
    "asr_gru_fwd_state": (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_void_p]),
    "asr_gru_bwd_state": (c_int, [c_void_p] * 13 + [c_int] * 4 + [c_void_p, c_void_p]),
    "asr_sru_ws_bytes": (c_size_t, [c_int] * 3),
    "asr_sru_fwd": (c_int, [c_void_p] * 9 + [c_int] * 4 + [c_void_p, c_size_t]),
    "asr_sru_bwd": (c_int, [c_void_p] * 13 + [c_int] * 4 + [c_void_p, c_size_t]),
    "asr_sru_combine": (c_int, [c_void_p] * 5 + [c_longlong, c_int]),
    "asr_fill_f32": (c_int, [c_void_p, c_void_p, c_longlong, c_float]),
    "asr_sqnorm_acc": (c_int, [c_void_p, c_void_p, c_longlong, c_void_p]),
    "asr_clip_decay_sgd": (c_int, [c_void_p] * 4 + [c_longlong, c_int] + [c_float] * 5 + [c_void_p]),
    "asr_clip_decay_adam": (c_int, [c_void_p] * 5 + [c_longlong] + [c_float] * 7 + [c_void_p, c_int]),
    "asr_sqnorm_partials_count": (c_int, [c_longlong]),
    "asr_gather_abort": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "asr_step_control": (c_int, [c_void_p, c_void_p, c_longlong] + [c_void_p] * 3 + [c_float] * 5 + [c_void_p] * 2 + [c_int]),
    "asr_step_control_scaled": (c_int, [c_void_p, c_void_p, c_longlong] + [c_void_p] * 3 + [c_float] * 5 + [c_void_p] * 2 + [c_int, c_void_p]),
    "asr_adam_ctl": (c_int, [c_void_p] * 5 + [c_longlong] + [c_float] * 4 + [c_void_p]),
    "asr_crelu_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int]),
    "asr_crelu_bwd": (c_int, [c_void_p] * 4 + [c_longlong, c_int]),
    "asr_softmax_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int]),
    "asr_softmax_bwd": (c_int, [c_void_p] * 4 + [c_longlong, c_int, c_int]),
    "asr_avgpool_h_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int, c_int]),
    "asr_avgpool_h_bwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int, c_int]),
    "asr_unpool_h_fwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int, c_int, c_int]),
    "asr_unpool_h_bwd": (c_int, [c_void_p] * 3 + [c_longlong, c_int, c_int, c_int, c_int]),
    "asr_gaussian_noise": (c_int, [c_void_p] * 3 + [c_longlong, c_float, ctypes.c_uint]),
    "asr_maxpool_h_indexes": (c_int, [c_void_p, c_void_p, c_void_p, c_longlong, c_int, c_int, c_int]),
    "asr_upsample_h_fwd": (c_int, [c_void_p] * 4 + [c_longlong] + [c_int] * 4),
    "asr_upsample_h_bwd": (c_int, [c_void_p] * 4 + [c_longlong] + [c_int] * 4),
    "asr_spp_bins": (c_int, [c_int]),
    "asr_spp_fwd": (c_int, [c_void_p] * 4 + [c_int] * 5),
    "asr_spp_bwd": (c_int, [c_void_p] * 4 + [c_int] * 5),
    "asr_sgd_ctl": (c_int, [c_void_p] * 4 + [c_longlong, c_int] + [c_float] * 3 + [c_void_p]),
}

_ERRORS = {-1: "bad argument", -2: "workspace too small", -3: "unsupported shape", -4: "kernel launch failed"}


class AsrHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise AsrHipError("%s not found: build it with `make -C %s` (hipcc, gfx950). There is no CPU fallback."
                              % (LIB_PATH, os.path.dirname(LIB_PATH)))
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)       # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise AsrHipError("%s failed: %s (code %d)" % (what, _ERRORS.get(rc, "unknown error"), rc))


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  The tensor must be contiguous and on the GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise AsrHipError("libasr_hip operates on GPU tensors only (got %s); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise AsrHipError("non-contiguous tensor passed to libasr_hip")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream
