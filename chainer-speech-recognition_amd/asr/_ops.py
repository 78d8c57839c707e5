"""Thin, autograd-free Python faces of the C ABI (include/asr_hip.h): shape checks + pointer plumbing only.

Everything numeric happens inside libasr_hip.so.  torch is used for allocation and stream handles.
"""
import os

import torch

from . import _lib
from ._lib import check, ptr, stream

BF16 = _lib.act_dtype()      # the 16-bit activation format of the loaded library: bfloat16 (default build) or float16 (ASR_ACT=f16)
F32 = torch.float32


def _is_bf16(t):
    if t.dtype == BF16:
        return 1
    if t.dtype == F32:
        return 0
    raise TypeError("expected float32 or bfloat16, got %s" % t.dtype)


def gemm_nt(a, b, bias=None, out_dtype=BF16, out=None):
    """C[M,N] = a[M,K] @ b[N,K]^T (+ bias).  a, b bf16 with unit inner stride."""
    assert a.dtype == BF16 and b.dtype == BF16 and a.dim() == 2 and b.dim() == 2
    assert a.stride(1) == 1 and b.stride(1) == 1 and a.shape[1] == b.shape[1]
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(1) == 1
    rc = _lib.lib().asr_gemm_nt(stream(), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                                out.stride(0), ptr(bias), M, N, K, _is_bf16(out))
    check(rc, "asr_gemm_nt")
    return out


def gemm_nt_8ph(a, b, bias=None, out_dtype=BF16, out=None):
    """gemm_nt on the 256 x 256 tile / eight-wave kernel only (asr_gemm_nt_8ph: tests and timings; asr_gemm_nt routes to it by itself);
    raises when the product does not qualify (K % 64, N % 4, alignment)"""
    assert a.dtype == BF16 and b.dtype == BF16 and a.dim() == 2 and b.dim() == 2
    assert a.stride(1) == 1 and b.stride(1) == 1 and a.shape[1] == b.shape[1]
    M, K = a.shape
    N = b.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(1) == 1
    rc = _lib.lib().asr_gemm_nt_8ph(stream(), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(),
                                    out.stride(0), ptr(bias), M, N, K, _is_bf16(out))
    check(rc, "asr_gemm_nt_8ph")
    return out


def gemm_tn_acc(a, b, c):
    """c[M,N] += a[K,M]^T @ b[K,N]; a, b bf16 row-major (unit inner stride), c f32."""
    assert a.dtype == BF16 and b.dtype == BF16 and c.dtype == F32
    assert a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1 and a.shape[0] == b.shape[0]
    K, M = a.shape
    N = b.shape[1]
    assert c.shape == (M, N)
    rc = _lib.lib().asr_gemm_tn_acc(stream(), a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), c.data_ptr(),
                                    c.stride(0), M, N, K)
    check(rc, "asr_gemm_tn_acc")
    return c


TN_GROUP = [_lib.debug_flag("tn_group", 1) != 0]     # ASR_DEBUG tn_group=0: one launch per product (comparison)


def gemm_tn_acc_group(products):
    """c += a^T @ b for up to four (a, b, c) triples in one launch (gemm_tn_acc's operand rules for each)."""
    n = len(products)
    assert 1 <= n <= 4
    if not TN_GROUP[0]:
        for a, b, c in products:
            gemm_tn_acc(a, b, c)
        return
    for a, b, c in products:
        assert a.dtype == BF16 and b.dtype == BF16 and c.dtype == F32
        assert a.stride(1) == 1 and b.stride(1) == 1 and c.stride(1) == 1 and a.shape[0] == b.shape[0]
        assert c.shape == (a.shape[1], b.shape[1])
    import ctypes
    ptrs = lambda k: (ctypes.c_void_p * n)(*[p[k].data_ptr() for p in products])
    ints = lambda f: (ctypes.c_int * n)(*[f(p) for p in products])
    rc = _lib.lib().asr_gemm_tn_acc_group(stream(), n, ptrs(0), ints(lambda p: p[0].stride(0)), ptrs(1),
                                          ints(lambda p: p[1].stride(0)), ptrs(2), ints(lambda p: p[2].stride(0)),
                                          ints(lambda p: p[0].shape[1]), ints(lambda p: p[1].shape[1]),
                                          ints(lambda p: p[0].shape[0]))
    check(rc, "asr_gemm_tn_acc_group")


def gemm_tn_acc_group_8ph(products):
    """gemm_tn_acc_group on the 256 x 256 tile / eight-wave kernel only (asr_gemm_tn_acc_group_8ph: tests and timings)"""
    n = len(products)
    assert 1 <= n <= 4
    import ctypes
    ptrs = lambda k: (ctypes.c_void_p * n)(*[p[k].data_ptr() for p in products])
    ints = lambda f: (ctypes.c_int * n)(*[f(p) for p in products])
    rc = _lib.lib().asr_gemm_tn_acc_group_8ph(stream(), n, ptrs(0), ints(lambda p: p[0].stride(0)), ptrs(1),
                                              ints(lambda p: p[1].stride(0)), ptrs(2), ints(lambda p: p[2].stride(0)),
                                              ints(lambda p: p[0].shape[1]), ints(lambda p: p[1].shape[1]),
                                              ints(lambda p: p[0].shape[0]))
    check(rc, "asr_gemm_tn_acc_group_8ph")


def cast_bf16(src, transpose=False, out=None):
    """f32 (rows, cols) -> bf16 copy, optionally transposed; `out`: a contiguous bf16 tensor of the result's size."""
    assert src.dtype == F32 and src.is_contiguous()
    s2 = src.reshape(src.shape[0], -1) if src.dim() != 2 else src
    rows, cols = s2.shape
    dst = out if out is not None else torch.empty((cols, rows) if transpose else (rows, cols), dtype=BF16, device=src.device)
    assert dst.dtype == BF16 and dst.is_contiguous() and dst.numel() == rows * cols
    check(_lib.lib().asr_cast_bf16(stream(), ptr(s2), ptr(dst), rows, cols, int(transpose)), "asr_cast_bf16")
    return dst


def bf16_to_f32(src):
    dst = torch.empty(src.shape, dtype=F32, device=src.device)
    check(_lib.lib().asr_bf16_to_f32(stream(), ptr(src.contiguous()), ptr(dst), src.numel()), "asr_bf16_to_f32")
    return dst


def permute4(src, shape, strides, out_dtype):
    """dense (d0,d1,d2,d3) tensor of `out_dtype` gathered from `src` storage with the given element strides."""
    dst = torch.empty(shape, dtype=out_dtype, device=src.device)
    rc = _lib.lib().asr_permute4(stream(), src.data_ptr(), _is_bf16(src), ptr(dst), _is_bf16(dst), *shape, *strides)
    check(rc, "asr_permute4")
    return dst


def im2col(x, strides_tbhc, T, B, Hin, Cin, KH, KW, pad_h, pad_t=None, Tout=None):
    """pad_t defaults to KW-1 and Tout to T: the causal convolution."""
    pad_t = KW - 1 if pad_t is None else pad_t
    Tout = T if Tout is None else Tout
    Kp = (KH * KW * Cin + 7) // 8 * 8
    Hout = Hin + 2 * pad_h - KH + 1
    col = torch.empty((Tout * B * Hout, Kp), dtype=BF16, device=x.device)
    rc = _lib.lib().asr_im2col(stream(), x.data_ptr(), _is_bf16(x), *strides_tbhc, T, B, Hin, Cin, KH, KW, pad_h, pad_t,
                               Tout, Kp, ptr(col))
    check(rc, "asr_im2col")
    return col


def col2im(dcol, T, B, Hin, Cin, KH, KW, pad_h, pad_t=None, Tout=None):
    pad_t = KW - 1 if pad_t is None else pad_t
    Tout = T if Tout is None else Tout
    Kp = dcol.shape[1]
    dx = torch.empty((T, B, Hin, Cin), dtype=BF16, device=dcol.device)
    check(_lib.lib().asr_col2im(stream(), ptr(dcol), T, B, Hin, Cin, KH, KW, pad_h, pad_t, Tout, Kp, ptr(dx)), "asr_col2im")
    return dx


def conv_weight_pack(W, transpose=False, Kp=None):
    """(Co, Ci, kh, kw) f32 -> bf16 (Co, Kp) [or (Kp, Co)], k = (kh, kw, ci), Kp = K rounded up to 8 (or as given), zero padded."""
    assert W.dtype == F32 and W.is_contiguous() and W.dim() == 4
    Co, Ci, KH, KW = W.shape
    Kp = (KH * KW * Ci + 7) // 8 * 8 if Kp is None else int(Kp)
    dst = torch.empty((Kp, Co) if transpose else (Co, Kp), dtype=BF16, device=W.device)
    check(_lib.lib().asr_conv_weight_pack(stream(), ptr(W), ptr(dst), Co, Ci, KH, KW, Kp, int(transpose)), "asr_conv_weight_pack")
    return dst


def conv_weight_grad_unpack(scratch, gW, channel_pitch=0):
    """gW (Co, Ci, KH, KW) += scratch (Co, Kp) with k = (kh, kw, c), c < channel_pitch (0: Ci); a scratch of shape (copies, Co, Kp)
    (per-XCD copies, conv_tn_copies) is summed over its first axis on the way"""
    Co, Ci, KH, KW = gW.shape
    assert scratch.dtype == F32 and scratch.is_contiguous() and gW.is_contiguous()
    copies = scratch.shape[0] if scratch.dim() == 3 else 1
    check(_lib.lib().asr_conv_weight_grad_unpack_copies(stream(), ptr(scratch), copies, ptr(gW), Co, Ci, KH, KW, scratch.shape[-1],
                                                        int(channel_pitch)), "asr_conv_weight_grad_unpack_copies")


def conv_tn_copies(Co, Cs, KH, KW):
    """8 if the weight-gradient kernel wants one scratch copy per XCD for this layer (few output tiles), else 1"""
    return int(_lib.lib().asr_conv_tn_copies(int(Co), int(Cs), int(KH), int(KW)))


def conv_tn_acc(g2, x, scratch, KH, KW, pad_h, pad_t, Tr, Hr):
    """scratch (Co, KH*KW*Cs) f32 += weight gradient of the convolution that maps x (Ts, B, Hs, Cs) bf16 to (Tr, B, Hr, Co),
    g2 (Tr*B*Hr, Co) bf16 its output gradient: the implicit form of gemm_tn_acc(g2, im2col(x), scratch).  scratch may be
    (8, Co, KH*KW*Cs) where conv_tn_copies says so: every XCD adds into its own copy."""
    Ts, B, Hs, Cs = x.shape
    Co = g2.shape[1]
    assert x.dtype == BF16 and g2.dtype == BF16 and scratch.dtype == F32 and g2.shape[0] == Tr * B * Hr
    copies = scratch.shape[0] if scratch.dim() == 3 else 1
    assert tuple(scratch.shape[-2:]) == (Co, KH * KW * Cs) and scratch.is_contiguous() and copies in (1, 8)
    rc = _lib.lib().asr_conv_tn_acc_copies(stream(), ptr(g2), g2.stride(0), ptr(x), ptr(scratch), scratch.shape[-1], copies, Co, Ts, B,
                                           Hs, Cs, KH, KW, pad_h, pad_t, Tr, Hr)
    check(rc, "asr_conv_tn_acc_copies")


def conv_tn_acc_8ph(g2, x, scratch, KH, KW, pad_h, pad_t, Tr, Hr):
    """conv_tn_acc on the eight-wave kernel only (asr_conv_tn_acc_8ph: tests and timings; asr_conv_tn_acc dispatches by itself)"""
    Ts, B, Hs, Cs = x.shape
    Co = g2.shape[1]
    assert tuple(scratch.shape) == (Co, KH * KW * Cs) and scratch.is_contiguous()
    rc = _lib.lib().asr_conv_tn_acc_8ph(stream(), ptr(g2), g2.stride(0), ptr(x), ptr(scratch), scratch.shape[-1], Co, Ts, B, Hs, Cs, KH, KW,
                                        pad_h, pad_t, Tr, Hr)
    check(rc, "asr_conv_tn_acc_8ph")


ACT_KINDS = {"relu": 0, "clipped_relu": 1, "leaky_relu": 2, "elu": 3, "sigmoid": 4, "tanh": 5, "hard_sigmoid": 6,
             "softplus": 7}


def activation_fwd(x, kind, alpha=0.0):
    assert x.dtype == BF16 and x.is_contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().asr_activation_fwd(stream(), ptr(x), ptr(y), x.numel(), ACT_KINDS[kind], float(alpha)), "asr_activation_fwd")
    return y


def activation_bwd(x, dy, kind, alpha=0.0):
    dx = torch.empty_like(x)
    check(_lib.lib().asr_activation_bwd(stream(), ptr(x), ptr(dy.contiguous()), ptr(dx), x.numel(), ACT_KINDS[kind],
                                        float(alpha)), "asr_activation_bwd")
    return dx


def glu_fwd(x):
    """x (..., 2C) bf16 -> (..., C): A * sigmoid(B)."""
    assert x.dtype == BF16 and x.is_contiguous() and x.shape[-1] % 2 == 0
    C = x.shape[-1] // 2
    y = torch.empty(x.shape[:-1] + (C,), dtype=BF16, device=x.device)
    check(_lib.lib().asr_glu_fwd(stream(), ptr(x), ptr(y), x.numel() // (2 * C), C), "asr_glu_fwd")
    return y


def glu_bwd(x, dy):
    C = x.shape[-1] // 2
    dx = torch.empty_like(x)
    check(_lib.lib().asr_glu_bwd(stream(), ptr(x), ptr(dy.contiguous()), ptr(dx), x.numel() // (2 * C), C), "asr_glu_bwd")
    return dx


def dropout(x, ratio, seed):
    assert x.dtype == BF16
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().asr_dropout(stream(), ptr(x), ptr(y), x.numel(), float(ratio), int(seed) & 0xffffffff), "asr_dropout")
    return y


def maxout2_fwd(x):
    assert x.dtype == BF16 and x.is_contiguous() and x.shape[-1] % 2 == 0
    y = torch.empty(x.shape[:-1] + (x.shape[-1] // 2,), dtype=BF16, device=x.device)
    check(_lib.lib().asr_maxout2_fwd(stream(), ptr(x), ptr(y), y.numel()), "asr_maxout2_fwd")
    return y


def maxout2_bwd(x, dy):
    dx = torch.empty_like(x)
    check(_lib.lib().asr_maxout2_bwd(stream(), ptr(x), ptr(dy.contiguous()), ptr(dx), dy.numel()), "asr_maxout2_bwd")
    return dx


def pooled_height(Hin, k):
    return 1 if Hin <= k else -(-(Hin - k) // k) + 1


def maxout2_pool_ok(x):
    return x.dtype == BF16 and x.dim() == 4 and x.is_contiguous() and x.shape[3] % 16 == 0


def maxout2_pool_fwd(x, k):
    """x (T, B, H, 2C) bf16 -> (T, B, ceil(H / k), C): Maxout(2) then MaxPooling2D((k, 1)) in one pass."""
    T, B, H, C2 = x.shape
    y = torch.empty((T, B, pooled_height(H, k), C2 // 2), dtype=BF16, device=x.device)
    check(_lib.lib().asr_maxout2_pool_fwd(stream(), ptr(x), ptr(y), T * B, H, C2 // 2, k), "asr_maxout2_pool_fwd")
    return y


def maxout2_pool_bwd(x, dy, k, db=None):
    """dx of maxout(2) + max pooling over (k, 1); db (2 C floats, optional) += column sums of dx (bias gradient of the producer)"""
    T, B, H, C2 = x.shape
    dx = torch.empty_like(x)
    if db is not None:
        assert db.dtype == F32 and db.numel() == C2 and db.is_contiguous()
    check(_lib.lib().asr_maxout2_pool_bwd_db(stream(), ptr(x), ptr(dy.contiguous()), ptr(dx), ptr(db), T * B, H, C2 // 2, k),
          "asr_maxout2_pool_bwd_db")
    return dx


def maxout2_pool_bwd_db_ok(C):
    return bool(_lib.lib().asr_maxout2_pool_bwd_db_ok(int(C)))


def maxpool_h_fwd(x, k):
    """x (T, B, H, C) bf16 -> (T, B, Hout, C)."""
    assert x.dtype == BF16 and x.is_contiguous() and x.dim() == 4
    T, B, H, C = x.shape
    y = torch.empty((T, B, pooled_height(H, k), C), dtype=BF16, device=x.device)
    check(_lib.lib().asr_maxpool_h_fwd(stream(), ptr(x), ptr(y), T * B, H, C, k), "asr_maxpool_h_fwd")
    return y


def maxpool_h_bwd(x, dy, k):
    T, B, H, C = x.shape
    dx = torch.empty_like(x)
    check(_lib.lib().asr_maxpool_h_bwd(stream(), ptr(x), ptr(dy.contiguous()), ptr(dx), T * B, H, C, k), "asr_maxpool_h_bwd")
    return dx


def add_bf16(a, b):
    assert a.dtype == BF16 and b.dtype == BF16 and a.shape == b.shape
    y = torch.empty_like(a, memory_format=torch.contiguous_format)
    check(_lib.lib().asr_add_bf16(stream(), ptr(a.contiguous()), ptr(b.contiguous()), ptr(y), y.numel()), "asr_add_bf16")
    return y


def colsum_acc(x, out):
    """out[c] += sum_r x[r][c]; x 2-d (unit inner stride), out f32."""
    assert x.dim() == 2 and x.stride(1) == 1 and out.dtype == F32
    rc = _lib.lib().asr_colsum_acc(stream(), x.data_ptr(), _is_bf16(x), x.shape[0], x.shape[1], x.stride(0), ptr(out))
    check(rc, "asr_colsum_acc")
    return out


def layernorm_fwd(x, gamma, beta, C, out_dtype, want_lse=False):
    """x (rows, D) f32/bf16; statistics per row; channel = index % C.  want_lse (float32 rows normalised over their whole width only):
    also the log-sum-exp of every output row, formed while the row is in registers -> (y, mean, rstd, lse)"""
    assert x.dim() == 2 and x.is_contiguous()
    rows, D = x.shape
    y = torch.empty((rows, D), dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=F32, device=x.device)
    rstd = torch.empty(rows, dtype=F32, device=x.device)
    if want_lse:
        assert x.dtype == F32 and out_dtype == F32 and _lib.lib().asr_layernorm_fwd_lse_ok(D, C)
        lse = torch.empty(rows, dtype=F32, device=x.device)
        rc = _lib.lib().asr_layernorm_fwd_lse(stream(), ptr(x), ptr(y), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(lse), rows, D)
        check(rc, "asr_layernorm_fwd_lse")
        return y, mean, rstd, lse
    rc = _lib.lib().asr_layernorm_fwd(stream(), ptr(x), _is_bf16(x), ptr(y), _is_bf16(y), ptr(gamma), ptr(beta),
                                      ptr(mean), ptr(rstd), rows, D, C)
    check(rc, "asr_layernorm_fwd")
    return y, mean, rstd


def layernorm_bwd(x, dy, gamma, mean, rstd, C, dx_dtype, dgamma=None, dbeta=None, need_dx=True):
    """dx (of dx_dtype) and dgamma += / dbeta +=.  float32 x and dy (the logits path) take the one-sweep kernel
    (asr_layernorm_bwd_rows: dx and the column sums in one pass), everything else asr_layernorm_bwd."""
    rows, D = x.shape
    dy = dy.contiguous()
    dx = torch.empty((rows, D), dtype=dx_dtype, device=x.device) if need_dx else None
    lib = _lib.lib()
    if x.dtype == F32 and dy.dtype == F32 and C % 4 == 0:
        nbytes = lib.asr_layernorm_bwd_rows_ws_bytes(rows, D)
        if nbytes > 0:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if dgamma is not None else None
            rc = lib.asr_layernorm_bwd_rows(stream(), ptr(x), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx),
                                            _is_bf16(dx) if need_dx else 0, ptr(dgamma), ptr(dbeta), rows, D, C, ptr(ws), nbytes)
            check(rc, "asr_layernorm_bwd_rows")
            return dx
    rc = lib.asr_layernorm_bwd(stream(), ptr(x), _is_bf16(x), ptr(dy), _is_bf16(dy), ptr(gamma), ptr(mean),
                               ptr(rstd), ptr(dx), _is_bf16(dx) if need_dx else 0, ptr(dgamma), ptr(dbeta),
                               rows, D, C)
    check(rc, "asr_layernorm_bwd")
    return dx


CALLS = {}          # call counters of a few entry points (tests check which path ran)


def layernorm_ctc_bwd(x, gamma, beta, mean, rstd, T, B, dx_dtype, dgamma, dbeta, need_dx, recipes, dxsum=None):
    """asr_layernorm_ctc_bwd: `recipes` = one or two dicts left by CTC-family losses on the normalised rows (asr/loss/ctc.py):
    ws, Lmax, gram, x_len, gy, gy_per_utt, scale"""
    rows, V = x.shape
    assert x.dtype == F32 and x.is_contiguous() and rows == T * B and 1 <= len(recipes) <= 2
    CALLS["layernorm_ctc_bwd"] = CALLS.get("layernorm_ctc_bwd", 0) + 1
    lib = _lib.lib()
    dx = torch.empty((rows, V), dtype=dx_dtype, device=x.device) if need_dx else None
    nbytes = lib.asr_layernorm_ctc_bwd_ws_bytes(T, B, V)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device) if dgamma is not None else None
    args = []
    for r in (recipes + recipes)[:2]:
        args += [ptr(r["ws"]), int(r["Lmax"]), int(r["gram"]), ptr(r["x_len"]), ptr(r["gy"]), int(r["gy_per_utt"]), float(r["scale"])]
    rc = lib.asr_layernorm_ctc_bwd(stream(), ptr(x), ptr(gamma), ptr(beta), ptr(mean), ptr(rstd), ptr(dx), _is_bf16(dx) if need_dx else 0,
                                   ptr(dgamma), ptr(dbeta), T, B, V, ptr(ws), nbytes, len(recipes), *args, ptr(dxsum))
    check(rc, "asr_layernorm_ctc_bwd")
    return dx


GRU_MODE = [int(os.environ.get("ASR_GRU_MODE", "0"))]      # (ASR_GRU_MODE: rehearsals with several processes on one GPU use 1)  asr_hip.h: 0 automatic, 1 per-step launches, 2 persistent placement-free, 4 local with flags, 7 forged placement, 8 local with polled payload (= 0), 9 / 10 partial-sum backward (asked for / forged placement)


_SYNC = {}          # (device index, stream) -> one reusable control buffer; its abort word (int 1023) is sticky
_STATUS = {}        # (device index, stream) -> [pinned host word, event of the copy in flight]


def _sync_buffer(dev, nbytes):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _SYNC.get(key)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.zeros(max(nbytes // 4 + 1, 2048), dtype=torch.int32, device=dev)
        _SYNC[key] = buf
    return buf


def gru_poll_status():
    """Called once per step (optimizer.update): looks, without synchronising, at the abort words copied out during the
    previous step and raises if a persistent GRU launch gave up; then queues this step's copy."""
    for key, buf in list(_SYNC.items()):         # one control buffer per (device, stream) that launched a recurrence
        st = _STATUS.get(key)
        if st is None:
            st = _STATUS[key] = [torch.zeros(1, dtype=torch.int32).pin_memory(), None]
        if st[1] is not None and st[1].query() and int(st[0][0]) != 0:
            code = int(st[0][0])
            buf[1023:1024].zero_()
            st[0][0] = 0
            st[1] = None
            raise _lib.AsrHipError("a persistent GRU kernel gave up an in-launch wait (code %d): the results of that step are "
                                   "invalid" % code)
        if st[1] is None or st[1].query():
            st[0].copy_(buf[1023:1024], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st[1] = ev


def reset_poll_status():
    """forget the abort-word copies in flight (the optimiser has just reported and cleared the words)"""
    for st in _STATUS.values():
        st[0][0] = 0
        st[1] = None


# asr.parallel hangs two callables here during a data-parallel backward pass: "before" runs before a recurrence is
# queued (the launch stream then waits for every collective in flight), "after" right behind it (complete gradient slices
# are all-reduced in the gap that follows).  A persistent recurrence wants every CU; see asr/parallel.py.
RECURRENCE_HOOKS = {"before": None, "after": None}


def _hook(name):
    fn = RECURRENCE_HOOKS[name]
    if fn is not None:
        fn()


GRU_GI_BF16 = [True]        # write the input projections in bf16 where the recurrence kernel takes them (tests may switch it off)
# saved gates in IEEE half, blocked by workgroup, where the default kernel pair serves (asr_hip.h): a row's r | z | n | q of 16 units
# are ONE 128-B line instead of four 64-B pieces of four lines.  Measured at T=1000, B=32, H=512: forward 1.256 -> 1.210 us per step,
# backward 1.347 -> 1.329 (half gates in the plain [4][H] layout had bought nothing: the CU's memory queue beside the hand-off is
# bound by the NUMBER of requests).  ASR_DEBUG gru_gates_f16=0 keeps float32 gates.
GRU_GATES_F16 = [_lib.debug_flag("gru_gates_f16", 1) != 0]


def gru_gates_standard(gates, H):
    """the saved gates as float32 (T*B, ndir, 4, H) whatever form gru_fwd kept them in (tests, inspection): the half form is blocked
    by workgroup, (T*B, ndir, H/16, 4, 16) -- asr_hip.h"""
    if gates.dtype != torch.float16:
        return gates
    rows, ndir = gates.shape[0], gates.shape[1]
    return gates.reshape(rows, ndir, H // 16, 4, 16).permute(0, 1, 3, 2, 4).reshape(rows, ndir, 4, H).float()


def gru_gates_f16(T, B, H, ndir):
    """whether a recurrence of this shape keeps its saved gates in half precision (the rounding-matched oracle asks)"""
    return bool(GRU_GATES_F16[0] and _lib.lib().asr_gru_gates_f16_ok(T, B, H, ndir, GRU_MODE[0]))


def gru_gi_dtype(T, B, H, ndir):
    """dtype the input-projection GEMM should write for the recurrence that follows: bf16 when the kernel serving this
    shape streams it (half the projection's output bytes and of the recurrence's loader traffic), else float32"""
    if GRU_GI_BF16[0] and _lib.lib().asr_gru_fwd_accepts_bf16_gi(T, B, H, ndir, GRU_MODE[0]):
        return BF16
    return F32


def _len_i32(x_len, B, dev):
    if x_len is None:
        return None
    if x_len.dtype != torch.int32 or x_len.numel() != B or x_len.device != dev:
        raise ValueError("x_length must be %d int32 frame counts on %s" % (B, dev))
    return x_len.contiguous()


def gru_fwd(gi, whh16, bhh, T, B, H, ndir, x_len=None):
    """x_len (B) int32 or None: per-utterance lengths (asr_hip.h: the dead rows' update-gate columns of gi are overwritten)"""
    dev = gi.device
    x_len = _len_i32(x_len, B, dev)
    sync = _sync_buffer(dev, _lib.lib().asr_gru_sync_bytes(B, H, ndir))
    hseq = torch.empty((T * B, ndir * H), dtype=F32, device=dev)
    hseq16 = torch.empty((T * B, ndir * H), dtype=BF16, device=dev)
    f16 = bool(GRU_GATES_F16[0] and _lib.lib().asr_gru_gates_f16_ok(T, B, H, ndir, GRU_MODE[0]))
    gates = torch.empty((T * B, ndir, 4, H), dtype=torch.float16 if f16 else F32, device=dev)
    y = torch.empty((T * B, H), dtype=BF16, device=dev)
    rc = _lib.lib().asr_gru_fwd(stream(), ptr(gi), _is_bf16(gi), ptr(whh16), ptr(bhh), ptr(hseq), ptr(hseq16), ptr(gates), ptr(y),
                                T, B, H, ndir, ptr(sync), GRU_MODE[0], ptr(x_len), int(f16))
    check(rc, "asr_gru_fwd")
    LAST_SYNC[0] = sync
    return y, hseq, hseq16, gates


def gru_fwd_state(gi, whh16, bhh, hx, T, B, H, ndir, x_len=None):
    """the layer with an initial state hx (ndir, B, H) float32 (asr_gru_fwd_state: per-step kernels, float32 gi and gates)"""
    dev = gi.device
    assert gi.dtype == F32 and hx.dtype == F32 and hx.shape == (ndir, B, H) and hx.is_contiguous()
    x_len = _len_i32(x_len, B, dev)
    hseq = torch.empty((T * B, ndir * H), dtype=F32, device=dev)
    hseq16 = torch.empty((T * B, ndir * H), dtype=BF16, device=dev)
    gates = torch.empty((T * B, ndir, 4, H), dtype=F32, device=dev)
    y = torch.empty((T * B, H), dtype=BF16, device=dev)
    rc = _lib.lib().asr_gru_fwd_state(stream(), ptr(gi), ptr(whh16), ptr(bhh), ptr(hx), ptr(hseq), ptr(hseq16), ptr(gates), ptr(y),
                                      T, B, H, ndir, ptr(x_len))
    check(rc, "asr_gru_fwd_state")
    return y, hseq, hseq16, gates


def gru_bwd_state(dy, gates, hseq, hx, dhy, whhT16, T, B, H, ndir, db_ih=None, db_hh=None, x_len=None):
    """-> dgi, dgh (bf16 rows) and dhx (ndir, B, H) float32; dhy: gradient of the final state or None"""
    dev = dy.device
    x_len = _len_i32(x_len, B, dev)
    dy_ws = torch.empty((T * B, H), dtype=BF16, device=dev) if x_len is not None else None
    dgi = torch.empty((T * B, ndir * 3 * H), dtype=BF16, device=dev)
    dgh = torch.empty((T * B, ndir * 3 * H), dtype=BF16, device=dev)
    carry = torch.empty((ndir, B, H), dtype=F32, device=dev)
    dhx = torch.empty((ndir, B, H), dtype=F32, device=dev)
    rc = _lib.lib().asr_gru_bwd_state(stream(), ptr(dy.contiguous()), ptr(gates), ptr(hseq), ptr(hx), None if dhy is None else ptr(dhy.contiguous()),
                                      ptr(whhT16), ptr(dgi), ptr(dgh), ptr(carry), ptr(db_ih), ptr(db_hh), ptr(dhx), T, B, H, ndir,
                                      ptr(x_len), ptr(dy_ws))
    check(rc, "asr_gru_bwd_state")
    return dgi, dgh, dhx


LAST_SYNC = [None]


def gru_check_sync():
    """(debug / tests) synchronise and raise if the last persistent GRU launch abandoned an in-launch wait"""
    s = LAST_SYNC[0]
    if s is not None and int(s[1023:1024].cpu()[0]) != 0:
        s[1023:1024].zero_()
        raise _lib.AsrHipError("persistent GRU kernel timed out waiting for another workgroup")


def gru_check_all():
    """synchronising check of EVERY control buffer of this process: the evaluation path calls it where it reads results back
    anyway (asr.error.compute_minibatch_error), because forward-only loops never reach optimizer.update's polling"""
    bad = []
    for key, buf in list(_SYNC.items()):
        if int(buf[1023:1024].cpu()[0]) != 0:
            buf[1023:1024].zero_()
            bad.append(key)
    if bad:
        raise _lib.AsrHipError("a persistent GRU kernel gave up an in-launch wait: the outputs of that forward pass are invalid")


def gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, db_ih=None, db_hh=None, x_len=None):
    """db_ih / db_hh: (ndir * 3H) f32 buffers the bias gradients are accumulated into (optional)."""
    dev = dy.device
    x_len = _len_i32(x_len, B, dev)
    dy_ws = torch.empty((T * B, H), dtype=BF16, device=dev) if x_len is not None else None
    if gates.dtype == torch.float16 and (db_ih is None or db_hh is None):
        # half gates belong to the partial-sum backward kernel, which sums the bias gradients in registers: give it somewhere to put them
        scratch = torch.empty((2, ndir * 3 * H), dtype=F32, device=dev)
        fill_(scratch, 0.0)
        db_ih = scratch[0] if db_ih is None else db_ih
        db_hh = scratch[1] if db_hh is None else db_hh
    dgi = torch.empty((T * B, ndir * 3 * H), dtype=BF16, device=dev)
    dgh = torch.empty((T * B, ndir * 3 * H), dtype=BF16, device=dev)
    sync = _sync_buffer(dev, _lib.lib().asr_gru_sync_bytes(B, H, ndir))
    carry = torch.empty((ndir, B, H), dtype=F32, device=dev)
    _hook("before")
    rc = _lib.lib().asr_gru_bwd(stream(), ptr(dy.contiguous()), ptr(gates), ptr(hseq), ptr(whhT16), ptr(dgi), ptr(dgh),
                                ptr(carry), ptr(db_ih), ptr(db_hh), T, B, H, ndir, ptr(sync), GRU_MODE[0], ptr(x_len), ptr(dy_ws),
                                int(gates.dtype == torch.float16))
    check(rc, "asr_gru_bwd")
    _hook("after")
    LAST_SYNC[0] = sync
    return dgi, dgh


def fill_(t, value):
    assert t.dtype == F32 and t.is_contiguous()
    check(_lib.lib().asr_fill_f32(stream(), ptr(t), t.numel(), float(value)), "asr_fill_f32")
    return t


def sqnorm_acc(g, out):
    check(_lib.lib().asr_sqnorm_acc(stream(), ptr(g), g.numel(), ptr(out)), "asr_sqnorm_acc")
    return out


def clip_decay_adam(p, g, m, v, alpha, beta1, beta2, eps, weight_decay, clip, grad_scale, sqnorm, step):
    rc = _lib.lib().asr_clip_decay_adam(stream(), ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), alpha, beta1, beta2, eps,
                                        weight_decay, clip, grad_scale, ptr(sqnorm), int(step))
    check(rc, "asr_clip_decay_adam")


def abort_words():
    """the abort words of every control buffer a recurrence of this process has used (asr_hip.h: sync_ws int 1023)"""
    return [buf[1023:1024] for buf in _SYNC.values()]


_ABORT = {}         # device index -> [addresses of the abort words covered, device table of them, the ORed word]


def gather_abort(dev, poison=None):
    """ONE device word = the OR of every abort word on `dev` (any number of control buffers: a process that ran recurrences on three
    streams -- eval on the default stream, then two half-batch streams -- has three), queued on the current stream.  poison: an
    element of the local gradient buffer that gets a NaN when a word is raised (data parallelism: every rank then drops the step)."""
    words = [w for w in abort_words() if w.device == dev]
    ptrs = tuple(w.data_ptr() for w in words)           # (a control buffer that grew was replaced: same count, another address)
    st = _ABORT.get(dev.index)
    if st is None or st[0] != ptrs:
        table = torch.tensor(list(ptrs) or [0], dtype=torch.int64).to(dev)
        st = _ABORT[dev.index] = [ptrs, table, torch.zeros(1, dtype=torch.int32, device=dev) if st is None else st[2]]
    check(_lib.lib().asr_gather_abort(stream(), ptr(st[1]), len(words), ptr(st[2]), ptr(poison)), "asr_gather_abort")
    return st[2]


def clear_abort_words(dev):
    for w in abort_words():
        if w.device == dev:
            w.zero_()


def step_control(g, partials, clip, grad_scale, alpha, beta1, beta2, applied, ctl, any_abort=None, reserved_index=-1, loss_scale=None):
    """any_abort: the word gather_abort returned for this step (None: gathered here); reserved_index, loss_scale (device float[4] or
    None): see asr_hip.h"""
    if any_abort is None:
        any_abort = gather_abort(g.device)
    rc = _lib.lib().asr_step_control_scaled(stream(), ptr(g), g.numel(), ptr(partials), ptr(any_abort), None, float(clip),
                                            float(grad_scale), float(alpha), float(beta1), float(beta2), ptr(applied), ptr(ctl),
                                            int(reserved_index), None if loss_scale is None else ptr(loss_scale))
    check(rc, "asr_step_control_scaled")


def sqnorm_partials_count(n):
    return int(_lib.lib().asr_sqnorm_partials_count(int(n)))


def adam_ctl(p, g, m, v, beta1, beta2, eps, weight_decay, ctl):
    rc = _lib.lib().asr_adam_ctl(stream(), ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), beta1, beta2, eps, weight_decay, ptr(ctl))
    check(rc, "asr_adam_ctl")


def sgd_ctl(p, g, v, kind, lr, momentum, weight_decay, ctl):
    rc = _lib.lib().asr_sgd_ctl(stream(), ptr(p), ptr(g), ptr(v), p.numel(), int(kind), lr, momentum, weight_decay, ptr(ctl))
    check(rc, "asr_sgd_ctl")


SRU_CHUNKED = [True]        # (tests switch the chunked scans off to run the one-thread-per-column kernels)


def _sru_ws(T, B, D, dev):
    n = int(_lib.lib().asr_sru_ws_bytes(T, B, D)) if SRU_CHUNKED[0] else 0
    return (torch.empty(n, dtype=torch.uint8, device=dev), n) if n else (None, 0)


def sru_fwd(x, U, bias, c0, mask, use_tanh):
    """x (T, B, D) bf16, U (T*B, 3D) f32 -> H bf16, C f32, cT (B, D) f32."""
    T, B, D = x.shape
    H = torch.empty_like(x)
    C = torch.empty((T, B, D), dtype=F32, device=x.device)
    cT = torch.empty((B, D), dtype=F32, device=x.device)
    ws, n = _sru_ws(T, B, D, x.device)
    rc = _lib.lib().asr_sru_fwd(stream(), ptr(x), ptr(U), ptr(bias), ptr(c0), ptr(mask), ptr(H), ptr(C), ptr(cT), T, B, D,
                                int(bool(use_tanh)), ptr(ws), n)
    check(rc, "asr_sru_fwd")
    return H, C, cT


def sru_bwd(x, U, bias, C, c0, mask, gH, gcT, gbias, use_tanh):
    T, B, D = x.shape
    gU = torch.empty((T * B, 3 * D), dtype=BF16, device=x.device)
    gxh = torch.empty_like(x)
    gc0 = torch.empty((B, D), dtype=F32, device=x.device)
    ws, n = _sru_ws(T, B, D, x.device)
    rc = _lib.lib().asr_sru_bwd(stream(), ptr(x), ptr(U), ptr(bias), ptr(C), ptr(c0), ptr(mask), ptr(gH), ptr(gcT), ptr(gU),
                                ptr(gxh), ptr(gbias), ptr(gc0), T, B, D, int(bool(use_tanh)), ptr(ws), n)
    check(rc, "asr_sru_bwd")
    return gU, gxh, gc0


def sru_combine(a, b, mask):
    out = torch.empty_like(a)
    BD = a.shape[-2] * a.shape[-1]
    check(_lib.lib().asr_sru_combine(stream(), ptr(a), ptr(b), ptr(mask), ptr(out), a.numel(), BD), "asr_sru_combine")
    return out


def clip_decay_sgd(p, g, v, kind, lr, momentum, weight_decay, clip, grad_scale, sqnorm):
    rc = _lib.lib().asr_clip_decay_sgd(stream(), ptr(p), ptr(g), ptr(v), p.numel(), int(kind), lr, momentum, weight_decay,
                                       clip, grad_scale, ptr(sqnorm))
    check(rc, "asr_clip_decay_sgd")


def weightnorm_fwd(V, g):
    """V (Co, ...) f32, g (Co, 1, 1, 1) -> W (same shape as V) f32, norm (Co)."""
    Co = V.shape[0]
    K = V.numel() // Co
    W = torch.empty_like(V, memory_format=torch.contiguous_format)
    norm = torch.empty(Co, dtype=F32, device=V.device)
    check(_lib.lib().asr_weightnorm_fwd(stream(), ptr(V.contiguous()), ptr(g.contiguous()), ptr(W), ptr(norm), Co, K), "asr_weightnorm_fwd")
    return W, norm


def weightnorm_bwd(gW, V, g, norm, gV, gg):
    Co = V.shape[0]
    K = V.numel() // Co
    rc = _lib.lib().asr_weightnorm_bwd(stream(), ptr(gW), ptr(V.contiguous()), ptr(g.contiguous()), ptr(norm), ptr(gV), ptr(gg), Co, K)
    check(rc, "asr_weightnorm_bwd")


def channel_mean_std(x2):
    rows, C = x2.shape
    mean = torch.empty(C, dtype=F32, device=x2.device)
    std = torch.empty(C, dtype=F32, device=x2.device)
    check(_lib.lib().asr_channel_stats(stream(), ptr(x2), rows, C, ptr(mean), ptr(std)), "asr_channel_stats")
    return mean, std


def weightnorm_init(mean, std):
    g = torch.empty_like(mean)
    b = torch.empty_like(mean)
    check(_lib.lib().asr_weightnorm_init(stream(), ptr(mean), ptr(std), ptr(g), ptr(b), mean.numel()), "asr_weightnorm_init")
    return g, b


def channel_affine(x2, scale, shift):
    y = torch.empty(x2.shape, dtype=BF16, device=x2.device)
    check(_lib.lib().asr_channel_affine(stream(), ptr(x2), ptr(scale), ptr(shift), ptr(y), x2.numel(), x2.shape[1]), "asr_channel_affine")
    return y


# ------------------------------------------------------------------------------------------------ decode / CER / loader extras
I32 = torch.int32


def argmax_rows(logits):
    """(T, B, V) f32 logits -> (B, T) int32 greedy ids."""
    T, B, V = logits.shape
    ids = torch.empty((B, T), dtype=I32, device=logits.device)
    check(_lib.lib().asr_argmax_rows(stream(), ptr(logits), T, B, V, ptr(ids)), "asr_argmax_rows")
    return ids


def ctc_collapse(ids, lengths, blank, merge_repeats=True):
    """(B, T) int32 -> (compacted ids (B, T) padded with blank, lengths (B))."""
    B, T = ids.shape
    out = torch.empty_like(ids)
    out_len = torch.empty((B,), dtype=I32, device=ids.device)
    rc = _lib.lib().asr_ctc_collapse(stream(), ptr(ids), None if lengths is None else ptr(lengths), B, T, int(blank), int(merge_repeats),
                                     ptr(out), ptr(out_len))
    check(rc, "asr_ctc_collapse")
    return out, out_len


def edit_distance(ref, ref_len, hyp, hyp_len):
    """row-wise Levenshtein distance of two padded int32 id matrices -> (pairs,) int32."""
    pairs = ref.shape[0]
    dist = torch.empty((pairs,), dtype=I32, device=ref.device)
    rc = _lib.lib().asr_edit_distance(stream(), ptr(ref), ptr(ref_len), ref.shape[1], ptr(hyp), ptr(hyp_len), hyp.shape[1], pairs, ptr(dist))
    check(rc, "asr_edit_distance")
    return dist


def cmn_pspec(pspec, nframes):
    B, Fmax, nbins = pspec.shape
    check(_lib.lib().asr_cmn_pspec(stream(), ptr(pspec), ptr(nframes), B, Fmax, nbins), "asr_cmn_pspec")
    return pspec


def add_white_noise(signals, lengths, gain, seed):
    B, pitch = signals.shape
    check(_lib.lib().asr_add_white_noise(stream(), ptr(signals), ptr(lengths), pitch, B, ptr(gain), int(seed) & (2 ** 64 - 1)),
          "asr_add_white_noise")
    return signals


def running_stats_update(x, lengths, total_before, mean, nvar, mean32, std32):
    B, T = x.shape[0], x.shape[-1]
    CM = x.numel() // (B * T)
    rc = _lib.lib().asr_running_stats_update(stream(), ptr(x), ptr(lengths), B, CM, T, int(total_before), ptr(mean), ptr(nvar),
                                             ptr(mean32), ptr(std32))
    check(rc, "asr_running_stats_update")


def normalize_bcmt(x, mean32, std32):
    B, T = x.shape[0], x.shape[-1]
    CM = x.numel() // (B * T)
    check(_lib.lib().asr_normalize_bcmt(stream(), ptr(x), ptr(mean32), ptr(std32), B, CM, T), "asr_normalize_bcmt")
    return x


def augment_specgram(pspec, nframes_out, speed, ratio, Fmax_out):
    """(B, Fin, nbins) f32 power spectra -> (B, Fmax_out, nbins): asr/fft.py:21-50 with given float64 factors per utterance."""
    B, Fin, nbins = pspec.shape
    out = torch.empty((B, Fmax_out, nbins), dtype=F32, device=pspec.device)
    rc = _lib.lib().asr_augment_specgram(stream(), ptr(pspec), ptr(nframes_out), ptr(speed), ptr(ratio), B, Fin, Fmax_out, nbins, ptr(out))
    check(rc, "asr_augment_specgram")
    return out


def batchnorm_stats(x2, eps, decay, avg_mean=None, avg_var=None):
    """x2 (R, C) bf16 -> (mean, rstd) f32 (C); running averages updated in place when given."""
    R, C = x2.shape
    ws = torch.empty(2 * C, dtype=torch.float64, device=x2.device)
    mean = torch.empty(C, dtype=F32, device=x2.device)
    rstd = torch.empty(C, dtype=F32, device=x2.device)
    rc = _lib.lib().asr_batchnorm_stats(stream(), ptr(x2), R, C, float(eps), float(decay), ptr(ws), ptr(mean), ptr(rstd),
                                        ptr(avg_mean), ptr(avg_var))
    check(rc, "asr_batchnorm_stats")
    return mean, rstd


def rsqrt_eps(var, eps):
    out = torch.empty_like(var)
    check(_lib.lib().asr_rsqrt_eps(stream(), ptr(var), float(eps), ptr(out), var.numel()), "asr_rsqrt_eps")
    return out


def batchnorm_fwd(x2, mean, rstd, gamma, beta):
    R, C = x2.shape
    y = torch.empty_like(x2)
    check(_lib.lib().asr_batchnorm_fwd(stream(), ptr(x2), ptr(mean), ptr(rstd), ptr(gamma), ptr(beta), R, C, ptr(y)), "asr_batchnorm_fwd")
    return y


def batchnorm_bwd(x2, gy, mean, rstd, gamma, dgamma, dbeta, need_dx=True):
    R, C = x2.shape
    ws = torch.empty(2 * C, dtype=torch.float64, device=x2.device)
    dx = torch.empty_like(x2) if need_dx else None
    rc = _lib.lib().asr_batchnorm_bwd(stream(), ptr(x2), ptr(gy), ptr(mean), ptr(rstd), ptr(gamma), R, C, ptr(ws), ptr(dx), ptr(dgamma),
                                      ptr(dbeta))
    check(rc, "asr_batchnorm_bwd")
    return dx


def conv_implicit_ok(Cs, KH, KW):
    """the implicit-GEMM convolution wants whole 16-B chunks per tap and whole 32-wide K tiles"""
    return Cs % 8 == 0 and (KH * KW * Cs) % 32 == 0


def conv_nt(x, W2, bias, out_dtype, KH, KW, pad_h, pad_t, sgn, Tr, Hr):
    """x (Ts, B, Hs, Cs) bf16 contiguous; W2 (N, KH*KW*Cs) bf16 -> (Tr*B*Hr, N): asr_hip.h asr_conv_nt."""
    Ts, B, Hs, Cs = x.shape
    N = W2.shape[0]
    assert x.dtype == BF16 and x.is_contiguous() and W2.dtype == BF16 and W2.is_contiguous() and W2.shape[1] >= KH * KW * Cs
    out = torch.empty((Tr * B * Hr, N), dtype=out_dtype, device=x.device)
    rc = _lib.lib().asr_conv_nt(stream(), ptr(x), ptr(W2), W2.shape[1], ptr(out), _is_bf16(out), ptr(bias), Ts, B, Hs, Cs, KH, KW,
                                pad_h, pad_t, int(sgn), Tr, Hr, N)
    check(rc, "asr_conv_nt")
    return out


def conv_nt_8ph(x, W2, bias, out_dtype, KH, KW, pad_h, pad_t, sgn, Tr, Hr):
    """conv_nt on the eight-wave implicit-GEMM kernel only (csrc/gemm8.hip; Cs % 64 == 0): tests / tools, asr_conv_nt dispatches by itself"""
    Ts, B, Hs, Cs = x.shape
    N = W2.shape[0]
    out = torch.empty((Tr * B * Hr, N), dtype=out_dtype, device=x.device)
    rc = _lib.lib().asr_conv_nt_8ph(stream(), ptr(x), ptr(W2), W2.shape[1], ptr(out), _is_bf16(out), ptr(bias), Ts, B, Hs, Cs, KH, KW,
                                    pad_h, pad_t, int(sgn), Tr, Hr, N)
    check(rc, "asr_conv_nt_8ph")
    return out


def conv_nt_8pn(x, W2, bias, out_dtype, KH, KW, pad_h, pad_t, sgn, Tr, Hr):
    """conv_nt on the narrow eight-wave kernel only (N <= 128; csrc/gemm8.hip): tests / tools"""
    Ts, B, Hs, Cs = x.shape
    N = W2.shape[0]
    out = torch.empty((Tr * B * Hr, N), dtype=out_dtype, device=x.device)
    rc = _lib.lib().asr_conv_nt_8pn(stream(), ptr(x), ptr(W2), W2.shape[1], ptr(out), _is_bf16(out), ptr(bias), Ts, B, Hs, Cs, KH, KW,
                                    pad_h, pad_t, int(sgn), Tr, Hr, N)
    check(rc, "asr_conv_nt_8pn")
    return out


def conv_direct_nt(x, W2, bias, KH, KW, pad_h, pad_t, sgn, Tr, Hr):
    """conv_nt through the LDS-resident kernel (csrc/conv_direct.hip), bf16 out; raises where asr_conv_direct_ok says no (tests / tools:
    asr_conv_nt dispatches by itself)"""
    Ts, B, Hs, Cs = x.shape
    N = W2.shape[0]
    out = torch.empty((Tr * B * Hr, N), dtype=BF16, device=x.device)
    rc = _lib.lib().asr_conv_direct_nt(stream(), ptr(x), ptr(W2), W2.shape[1], ptr(out), ptr(bias), Ts, B, Hs, Cs, KH, KW, pad_h, pad_t,
                                       int(sgn), Tr, Hr, N)
    check(rc, "asr_conv_direct_nt")
    return out


def conv_mp_ok(Ci, KH, KW, Co, k):
    return bool(_lib.lib().asr_conv_mp_ok(int(Ci), int(KH), int(KW), int(Co), int(k)))


def conv_mp_fwd(x8, W2, bias, KH, KW, pad_h, pad_t, Tout, Hout, k):
    """first block in one pass (csrc/conv_first.hip): x8 (Ts, B, Hs, 8) bf16, W2 (Co, 128) bf16 -> pooled maxout of the convolution
    (Tout, B, Hp, Co / 2) bf16 and the winners' indices (same shape, uint8)"""
    Ts, B, Hs, Cs = x8.shape
    Co = W2.shape[0]
    assert Cs == 8 and x8.dtype == BF16 and x8.is_contiguous() and W2.dtype == BF16 and W2.is_contiguous() and W2.shape[1] == 128
    Hp = pooled_height(Hout, k)
    y = torch.empty((Tout, B, Hp, Co // 2), dtype=BF16, device=x8.device)
    idx = torch.empty((Tout, B, Hp, Co // 2), dtype=torch.uint8, device=x8.device)
    CALLS["conv_mp_fwd"] = CALLS.get("conv_mp_fwd", 0) + 1
    rc = _lib.lib().asr_conv_mp_fwd(stream(), ptr(x8), ptr(W2), 128, ptr(bias), ptr(y), ptr(idx), Ts, B, Hs, KH, KW, pad_h, pad_t, Tout,
                                    Hout, Co, k)
    check(rc, "asr_conv_mp_fwd")
    return y, idx


def conv_mp_bwd(gy, idx, x8, gW, gb, KH, KW, pad_h, pad_t, Hout, k):
    """gW (Co, Ci, KH, KW) f32 += weight gradient, gb (Co) f32 += bias gradient (None: none) of conv_mp_fwd"""
    Ts, B, Hs, _ = x8.shape
    Tout, _, Hp, Cp = gy.shape
    Co, Ci = gW.shape[0], gW.shape[1]
    assert gy.dtype == BF16 and gy.is_contiguous() and idx.shape == gy.shape and idx.dtype == torch.uint8 and idx.is_contiguous()
    assert gW.dtype == F32 and gW.is_contiguous() and Co == 2 * Cp and Hp == pooled_height(Hout, k)
    assert gb is None or (gb.dtype == F32 and gb.is_contiguous() and gb.numel() == Co)
    nbytes = _lib.lib().asr_conv_mp_bwd_workspace(Tout, B, Hout, Co, k)
    ws = torch.empty((nbytes // 4,), dtype=F32, device=gy.device)
    rc = _lib.lib().asr_conv_mp_bwd(stream(), ptr(gy), ptr(idx), ptr(x8), ptr(ws), ptr(gW), ptr(gb), Ts, B, Hs, Ci, KH, KW, pad_h, pad_t,
                                    Tout, Hout, Co, k)
    check(rc, "asr_conv_mp_bwd")


def conv_weight_pack_bwd(W):
    """(Co, Ci, kh, kw) f32 -> bf16 (Ci, kh*kw*Co), k = (kh, kw, co)."""
    Co, Ci, KH, KW = W.shape
    dst = torch.empty((Ci, KH * KW * Co), dtype=BF16, device=W.device)
    check(_lib.lib().asr_conv_weight_pack_bwd(stream(), ptr(W), ptr(dst), Co, Ci, KH, KW), "asr_conv_weight_pack_bwd")
    return dst


def pack_input_pad(x, strides_tbhc, T, B, H, C, Cpad):
    """any strided (T, B, H, C) view of x -> dense (T, B, H, Cpad) bf16 with zero channels behind C."""
    out = torch.empty((T, B, H, Cpad), dtype=BF16, device=x.device)
    rc = _lib.lib().asr_pack_input_pad(stream(), x.data_ptr(), _is_bf16(x), *strides_tbhc, T, B, H, C, Cpad, ptr(out))
    check(rc, "asr_pack_input_pad")
    return out


# ------------------------------------------------------------------------------------------------ the rest of asr.nn's function layers
def crelu_fwd(x):
    """x (..., C) bf16 contiguous -> (..., 2C): [relu(x) | relu(-x)] along the channels"""
    assert x.dtype == BF16 and x.is_contiguous()
    C = x.shape[-1]
    y = torch.empty(x.shape[:-1] + (2 * C,), dtype=BF16, device=x.device)
    check(_lib.lib().asr_crelu_fwd(stream(), ptr(x), ptr(y), x.numel() // C, C), "asr_crelu_fwd")
    return y


def crelu_bwd(x, dy):
    C = x.shape[-1]
    dx = torch.empty_like(x)
    check(_lib.lib().asr_crelu_bwd(stream(), ptr(x), ptr(dy.contiguous()), ptr(dx), x.numel() // C, C), "asr_crelu_bwd")
    return dx


def softmax_fwd(x, log_form):
    assert x.dtype == BF16 and x.is_contiguous()
    C = x.shape[-1]
    y = torch.empty_like(x)
    check(_lib.lib().asr_softmax_fwd(stream(), ptr(x), ptr(y), x.numel() // C, C, int(bool(log_form))), "asr_softmax_fwd")
    return y


def softmax_bwd(y, dy, log_form):
    C = y.shape[-1]
    dx = torch.empty_like(y)
    check(_lib.lib().asr_softmax_bwd(stream(), ptr(y), ptr(dy.contiguous()), ptr(dx), y.numel() // C, C, int(bool(log_form))), "asr_softmax_bwd")
    return dx


def avgpool_h_fwd(x, k):
    """x (T, B, H, C) bf16 -> (T, B, (H - k) // k + 1, C): mean over whole windows of k rows"""
    assert x.dtype == BF16 and x.is_contiguous() and x.dim() == 4
    T, B, H, C = x.shape
    y = torch.empty((T, B, (H - k) // k + 1, C), dtype=BF16, device=x.device)
    check(_lib.lib().asr_avgpool_h_fwd(stream(), ptr(x), ptr(y), T * B, H, C, k), "asr_avgpool_h_fwd")
    return y


def avgpool_h_bwd(dy, H, k):
    T, B, _, C = dy.shape
    dx = torch.empty((T, B, H, C), dtype=BF16, device=dy.device)
    check(_lib.lib().asr_avgpool_h_bwd(stream(), ptr(dy.contiguous()), ptr(dx), T * B, H, C, k), "asr_avgpool_h_bwd")
    return dx


def unpool_h_fwd(x, k, Hout):
    assert x.dtype == BF16 and x.is_contiguous() and x.dim() == 4
    T, B, H, C = x.shape
    y = torch.empty((T, B, Hout, C), dtype=BF16, device=x.device)
    check(_lib.lib().asr_unpool_h_fwd(stream(), ptr(x), ptr(y), T * B, H, Hout, C, k), "asr_unpool_h_fwd")
    return y


def unpool_h_bwd(dy, H, k):
    T, B, Hout, C = dy.shape
    dx = torch.empty((T, B, H, C), dtype=BF16, device=dy.device)
    check(_lib.lib().asr_unpool_h_bwd(stream(), ptr(dy.contiguous()), ptr(dx), T * B, H, Hout, C, k), "asr_unpool_h_bwd")
    return dx


def maxpool_h_indexes(x, k):
    """x (T, B, H, C) bf16 -> uint8 (T, B, Hout, C): row inside its window of the first maximum (what max_pooling_2d routes its gradient to)"""
    assert x.dtype == BF16 and x.is_contiguous() and x.dim() == 4
    T, B, H, C = x.shape
    idx = torch.empty((T, B, pooled_height(H, k), C), dtype=torch.uint8, device=x.device)
    check(_lib.lib().asr_maxpool_h_indexes(stream(), ptr(x), ptr(idx), T * B, H, C, k), "asr_maxpool_h_indexes")
    return idx


def upsample_h_fwd(x, idx, k, Hout):
    assert x.dtype == BF16 and x.is_contiguous() and idx.dtype == torch.uint8 and idx.is_contiguous() and idx.shape == x.shape
    T, B, H, C = x.shape
    y = torch.empty((T, B, Hout, C), dtype=BF16, device=x.device)
    check(_lib.lib().asr_upsample_h_fwd(stream(), ptr(x), ptr(idx), ptr(y), T * B, H, Hout, C, k), "asr_upsample_h_fwd")
    return y


def upsample_h_bwd(dy, idx, k):
    T, B, Hout, C = dy.shape
    H = idx.shape[2]
    dx = torch.empty((T, B, H, C), dtype=BF16, device=dy.device)
    check(_lib.lib().asr_upsample_h_bwd(stream(), ptr(dy.contiguous()), ptr(idx), ptr(dx), T * B, H, Hout, C, k), "asr_upsample_h_bwd")
    return dx


def spp_fwd(x, pyramid_height, want_pos):
    """x (T, B, H, C) bf16 -> y (B, bins, C) bf16 [, pos (B, bins, C) int32]"""
    assert x.dtype == BF16 and x.is_contiguous() and x.dim() == 4
    T, B, H, C = x.shape
    bins = _lib.lib().asr_spp_bins(int(pyramid_height))
    if bins <= 0:
        raise ValueError("pyramid_height must be in 1..8")
    y = torch.empty((B, bins, C), dtype=BF16, device=x.device)
    pos = torch.empty((B, bins, C), dtype=torch.int32, device=x.device) if want_pos else None
    check(_lib.lib().asr_spp_fwd(stream(), ptr(x), ptr(y), ptr(pos), T, B, H, C, int(pyramid_height)), "asr_spp_fwd")
    return y, pos


def spp_bwd(dy, pos, shape, pyramid_height):
    T, B, H, C = shape
    dx32 = torch.empty((T, B, H, C), dtype=F32, device=dy.device)
    fill_(dx32, 0.0)
    check(_lib.lib().asr_spp_bwd(stream(), ptr(dy.contiguous()), ptr(pos), ptr(dx32), T, B, H, C, int(pyramid_height)), "asr_spp_bwd")
    return cast_bf16(dx32.reshape(T * B * H, C)).reshape(T, B, H, C)


def gaussian_noise(x, std, seed):
    assert x.dtype == BF16
    x = x.contiguous()
    y = torch.empty_like(x)
    check(_lib.lib().asr_gaussian_noise(stream(), ptr(x), ptr(y), x.numel(), float(std), int(seed) & 0xffffffff), "asr_gaussian_noise")
    return y
