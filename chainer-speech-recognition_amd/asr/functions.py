"""Differentiable operators of the hot path: ``torch.autograd.Function`` shells around the C ABI.

This module plays the role ``chainer.functions`` plays for the reference (``import chainer.functions as F`` in
run/ctc/*/train.py and asr/model/*.py): a maintainer swaps that import for ``import asr.functions as F``.

Conventions
-----------
* Tensors keep the reference's LOGICAL shapes -- images (B, C, H, T), sequences (B, D, T) -- but live in HBM
  time-major / channel-last: the logical tensor is a permuted VIEW of a contiguous (T, B, H, C) or (T, B, D)
  bf16 buffer ("physical" form).  Array manipulations below are views; nothing here computes with torch.
* Parameter gradients are accumulated by the HIP kernels straight into ``param.grad`` (a slice of the optimiser's
  flat gradient buffer); the Functions return ``None`` for parameters.
"""
import math
import os

import torch

from . import _ops
from ._lib import debug_flag as _lib_debug
from .link import grad_buffer, grads_queued

BF16, F32 = _ops.BF16, torch.float32       # BF16: the library's 16-bit activation format (bfloat16, or float16 with ASR_ACT=f16)


# ---------------------------------------------------------------------------------------------- side stream
# Weight-gradient GEMMs do not feed the activation-gradient chain: they run on a second HIP stream, beside the next
# (earlier) layer's latency-bound GRU recurrence, which occupies only 64 of the 256 CUs.  The optimiser joins the stream.
# One side stream per launching stream (a data prefetcher or an evaluation pass may launch from streams of their own).
_SIDE = {"streams": {}, "origins": {}, "enabled": True, "dirty": set(), "keep": [], "keep_bytes": 0,
         "join_before_recurrence": _lib_debug("side_join", 0) != 0}
_KEEP_LIMIT = 512       # tensors held for a side stream by a caller that never joins (see _OnSide)
# ... and bytes: what a backward pass pins until the optimiser joins is every layer's gradient / column operand at once (hundreds of
# MB per convolution at T = 1000, B = 32: DESIGN.md 14.2); several passes without a join (gradient accumulation, bare backward loops) stop
# pinning beyond this and hand the older tensors to the allocator's own stream bookkeeping
_KEEP_BYTES_LIMIT = int(_lib_debug("side_keep_gb", 48)) << 30


def side_stream():
    if not _SIDE["enabled"]:
        return None
    cur = torch.cuda.current_stream()
    key = cur.cuda_stream
    st = _SIDE["streams"].get(key)
    if st is None:
        st = _SIDE["streams"][key] = torch.cuda.Stream(priority=_lib_debug("side_priority", 0))
        _SIDE["origins"][key] = cur
    return st


def side_streams():
    return list(_SIDE["streams"].values())


def join_side_stream():
    """called by the optimiser before it reads the gradients: the current stream -- and the stream every side stream was forked from --
    waits for the side streams; the tensors the side streams were reading are released only now (see _OnSide)"""
    cur = torch.cuda.current_stream()
    for key, st in _SIDE["streams"].items():
        cur.wait_stream(st)
        origin = _SIDE["origins"].get(key)
        if origin is not None and origin.cuda_stream != cur.cuda_stream:
            origin.wait_stream(st)
    _SIDE["dirty"].clear()
    release_side_keeps()


def release_side_keeps(to_allocator=False):
    """drop the references held for the side streams.  After a join they are stream-ordered already; `to_allocator`: no join happened (an
    exception left the backward pass half queued, or the limits were hit) -- the allocator's record_stream bookkeeping takes over."""
    keep = _SIDE["keep"]
    if to_allocator:
        for t, st in keep:
            t.record_stream(st)
    del keep[:]
    _SIDE["keep_bytes"] = 0


def _empty_chip_for_recurrence():
    """ASR_DEBUG side_join=1 (off by default): the launching stream waits for the side streams before it queues a recurrence.
    A persistent recurrence wants every CU to itself (132 KB of LDS per workgroup).  Queued while weight-gradient workgroups of
    the side stream are still resident it is dealt out over whatever CUs come free first: the workgroup -> XCD order the
    XCD-local hand-off relies on can be lost, the in-launch vote then falls back to the placement-free form and that launch runs
    1.6 .. 2.2 x longer (traced with tools/step_spread.py: 2.1 / 2.9 instead of 1.3 ms whenever a grouped product whose workgroups
    live 290 us had been queued BEHIND the input-gradient product).  With the products queued beside that product (see
    _GRU.backward) and 6 K splits the trace shows every recurrence at its normal length without the join, and the join itself
    costs 0.3 ms per step (four cross-stream waits, and the recurrence's ramp no longer overlaps the products' tail): 15.05-15.15
    against 14.73-14.80 ms.  Kept as a switch for set-ups where other kernels share the chip."""
    if not _SIDE["join_before_recurrence"] or not _SIDE["dirty"]:
        return
    cur = torch.cuda.current_stream()
    for st in _SIDE["dirty"]:
        cur.wait_stream(st)
    _SIDE["dirty"].clear()


class _OnSide(object):
    """with _OnSide(tensors...): kernels launched inside go to the side stream, after everything queued so far.

    `tensors` live on the launching stream and are read by the side stream.  They are kept ALIVE until join_side_stream() has made
    the launching stream wait for the side stream -- not handed to Tensor.record_stream().  record_stream defers the reuse of a block
    until an event on the side stream has completed, and the caching allocator looks at those events only when it is asked for memory:
    a host that queues a step in 4 ms while the device takes 14 - 38 ms runs many steps ahead, none of the events has completed when the
    next step asks, and every step in flight gets FRESH blocks from hipMalloc (measured with tools/cnn_step_times.py: 24 - 55 device
    allocations and 5.6 - 17.8 GB of growth during five free-running steps of each recipe, zero with a synchronisation per step; this
    is what made `extra_configs.cnn_wide8` 88.9 instead of 38.2 ms per step in one kept bench line -- hipMalloc calls inside a timed
    region of five steps).  Held references are stream-ordered by construction: after the join, the blocks return to
    the launching stream's pool behind the wait."""

    def __init__(self, *tensors):
        self.tensors = tensors
        self.side = side_stream()

    def __enter__(self):
        if self.side is None:
            return self
        self.side.wait_stream(torch.cuda.current_stream())
        _SIDE["dirty"].add(self.side)
        keep = _SIDE["keep"]
        if len(keep) > _KEEP_LIMIT or _SIDE["keep_bytes"] > _KEEP_BYTES_LIMIT:
            release_side_keeps(to_allocator=True)   # nobody joins (a loop of bare backward passes): the allocator's own bookkeeping
        for t in self.tensors:
            if t is not None:
                keep.append((t, self.side))
                _SIDE["keep_bytes"] += t.numel() * t.element_size()
        self.ctx = torch.cuda.stream(self.side)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.side is not None:
            self.ctx.__exit__(*exc)
        return False


# ---------------------------------------------------------------------------------------------- bf16 gradient hand-over
# A layer that writes float32 output (the logit projection) wants its incoming gradient in bf16, but autograd insists
# on a float32 gradient for a float32 tensor -- a 384 MB write plus a cast pass on the logits.  Such a producer hangs a
# mailbox on its graph node; a consumer that finds it (through pure view nodes) posts its bf16 gradient there and hands
# autograd an all-zero stride-0 token instead.  The producer adds mailbox + whatever autograd delivered, so the result is
# right with any number of consumers; with the usual single consumer the token is recognised and nothing is added.
class _GradMailbox(object):
    """`bias`: the producer's bias parameter; a consumer that forms the gradient of the producer's output inside its own sweep
    may add its column sums to that bias gradient there and then (`bias_done`), sparing the producer a pass over the rows"""
    __slots__ = ("ptr", "numel", "value", "bias", "bias_done")

    def __init__(self, y, bias=None):
        self.ptr, self.numel, self.value, self.bias, self.bias_done = y.data_ptr(), y.numel(), None, bias, False

    def post(self, g):
        self.value = g if self.value is None else _ops.add_bf16(self.value, g.reshape(self.value.shape))

    def take(self):
        v, self.value = self.value, None
        return v


class _BiasBox(object):
    """hung on a convolution's graph node: the consumer that forms the WHOLE gradient of the convolution's output (the fused
    maxout + pooling backward, when the layer stack guarantees it is the only consumer) adds its column sums to the bias gradient
    in the same pass and says so here"""
    __slots__ = ("ptr", "numel", "bias", "bias_done")

    def __init__(self, y, bias):
        self.ptr, self.numel, self.bias, self.bias_done = y.data_ptr(), y.numel(), bias, False


_ZERO = {}
_VIEW_NODES = ("ViewBackward", "ReshapeAliasBackward", "UnsafeViewBackward", "PermuteBackward", "TransposeBackward",
               "SqueezeBackward", "UnsqueezeBackward", "AliasBackward")


def _zero_token(shape, device):
    z = _ZERO.get(device)
    if z is None:
        z = _ZERO[device] = torch.zeros(1, dtype=F32, device=device)
    return z.reshape((1,) * len(shape)).expand(shape)


def _is_zero_token(t):
    z = _ZERO.get(t.device)
    return z is not None and t.data_ptr() == z.data_ptr()


def _producer_box(x2, attr="_asr_mailbox", hops=12, dtype=F32):
    """the box (attribute `attr`) of the node that produced the buffer `x2` is a contiguous re-view of, or None"""
    if x2.dtype != dtype or not x2.is_contiguous():
        return None
    node = x2.grad_fn
    for _ in range(hops):
        if node is None:
            return None
        box = getattr(node, attr, None)
        if box is not None:
            return box if (box.ptr == x2.data_ptr() and box.numel == x2.numel()) else None
        if not type(node).__name__.startswith(_VIEW_NODES) or len(node.next_functions) != 1:
            return None
        node = node.next_functions[0][0]
    return None


def _producer_mailbox(x2):
    return _producer_box(x2, "_asr_mailbox", 8)


# A CTC-family loss whose logits come straight out of a per-frame LayerNormalization does not write its gradient (384 MB of
# float32 at the BASELINE size, read back at once by the normalisation's backward): it leaves a RECIPE -- the workspace with
# alpha / beta / log-sum-exp of its forward pass, lengths, upstream gradient, scale -- in the box the normalisation hung on
# its graph node, and hands autograd a zero token (the first loss) or nothing (a second loss on the same logits: the joint
# Gram-CTC + CTC step of run/gram_ctc/cnn/train.py:163-167).  LayerNormalization's backward then forms (softmax - occupancy)
# in registers inside its own sweep (csrc/ctc_ln.hip).  Whatever else flows into the logits arrives through autograd as usual
# and is added by a second, plain backward sweep.
FUSE_CTC_INTO_LAYERNORM = [True]
LAST_FUSED_RECIPES = [0]        # number of loss recipes the last fused LayerNorm + CTC backward took (tests)


class _CtcBox(object):
    """`lse`: log-sum-exp of every (t, b) row of the logits (float32, T * B), or None"""
    __slots__ = ("ptr", "numel", "shape", "recipes", "lse")

    def __init__(self, y, T, B, V, lse=None):
        self.ptr, self.numel, self.shape, self.recipes, self.lse = y.data_ptr(), y.numel(), (T, B, V), [], lse

    def post(self, recipe):
        self.recipes.append(recipe)
        return len(self.recipes) == 1

    def take(self):
        r, self.recipes = self.recipes, []
        return r


def ctc_box_of(xs):
    """the recipe box of the LayerNormalization that produced the (T, B, V) logits `xs`, or None"""
    if not FUSE_CTC_INTO_LAYERNORM[0] or xs.dim() != 3:
        return None
    box = _producer_box(xs, "_asr_ctc_box")
    return box if (box is not None and box.shape == tuple(xs.shape)) else None


def _incoming_bf16(ctx, gy, width):
    """the gradient of a producer's output as contiguous bf16 rows (-1, width): autograd's part + the mailbox's.
    Returns (total, autograd_part): autograd_part is None when autograd delivered only the zero token, i.e. the mailbox's
    consumer was the only one -- a consumer that already added ITS column sums to the producer's bias gradient (bias_done) has
    not seen what other consumers sent through autograd, so the producer still owes the bias the column sums of that part."""
    box = getattr(ctx, "_asr_mailbox", None)
    extra = box.take() if box is not None else None
    if extra is not None and _is_zero_token(gy):
        return extra.reshape(-1, width), None
    gy = gy.contiguous()
    if gy.dtype != BF16:
        gy = _ops.cast_bf16(gy.reshape(-1, width))
    gy = gy.reshape(-1, width)
    if extra is not None:
        return _ops.add_bf16(gy, extra.reshape(-1, width)), gy
    return gy, gy


def _take_bias_done(ctx):
    """read AND reset the "a consumer has added its column sums to my bias gradient" flags of this node's boxes"""
    done = False
    for attr in ("_asr_mailbox", "_asr_biasbox"):
        box = getattr(ctx, attr, None)
        if box is not None and box.bias_done:
            done = True
            box.bias_done = False          # consumed: a second backward pass over the same graph starts afresh
    return done


# ---------------------------------------------------------------------------------------------- layout helpers
class _ToPhys(torch.autograd.Function):
    """any strided f32/bf16 tensor (logical order given by `perm`) -> contiguous bf16 in physical order."""

    @staticmethod
    def forward(ctx, x, perm):
        p = x.permute(*perm)
        shape = tuple(p.shape) + (1,) * (4 - p.dim())
        strides = tuple(p.stride()) + (0,) * (4 - p.dim())
        ctx.meta = (x.dtype, perm, tuple(p.shape))
        return _ops.permute4(x, shape, strides, BF16).reshape(p.shape)

    @staticmethod
    def backward(ctx, g):
        dtype, perm, pshape = ctx.meta
        inv = [0] * len(perm)
        for i, a in enumerate(perm):
            inv[a] = i
        g = g.contiguous()
        if dtype == BF16:
            return g.permute(*inv), None
        return _ops.bf16_to_f32(g).permute(*inv), None


def phys4(x):
    """logical (B, C, H, T) -> contiguous (T, B, H, C) bf16 (no copy when it already is one)."""
    if x.dim() != 4:
        raise ValueError("expected a (B, C, H, T) tensor, got shape %s" % (tuple(x.shape),))
    p = x.permute(3, 0, 2, 1)
    if p.dtype == BF16 and p.is_contiguous():
        return p
    return _ToPhys.apply(x, (3, 0, 2, 1))


def logical4(p):
    return p.permute(1, 3, 2, 0)


def phys3(x):
    """logical (B, D, T) -> contiguous (T, B, D) bf16."""
    if x.dim() != 3:
        raise ValueError("expected a (B, D, T) tensor, got shape %s" % (tuple(x.shape),))
    p = x.permute(2, 0, 1)
    if p.dtype == BF16 and p.is_contiguous():
        return p
    return _ToPhys.apply(x, (2, 0, 1))


def logical3(p):
    return p.permute(1, 2, 0)


def _phys_any(x):
    """(physical tensor, restore-to-logical fn) for 3-d or 4-d logical input."""
    if x.dim() == 4:
        return phys4(x), logical4
    if x.dim() == 3:
        return phys3(x), logical3
    if x.dim() == 2:
        if x.dtype == BF16 and x.is_contiguous():
            return x, (lambda p: p)
        return _ToPhys.apply(x, (0, 1)), (lambda p: p)
    raise ValueError("unsupported rank %d" % x.dim())


# ---------------------------------------------------------------------------------------------- convolution
class _Conv2D(torch.autograd.Function):
    """x: ANY strided 4-d (B, C, H, T) tensor (f32 or bf16); W (Co, Ci, kh, kw) f32 master; b (Co) or None.
    Output physical (Tout, B, Hout, Co) bf16 (or f32)."""

    @staticmethod
    def forward(ctx, x, W, b, w16, w16t, wbwd, w16p, pad_h, pad_t, causal, out_f32):
        B, Ci, Hin, T = x.shape
        Co, _, KH, KW = W.shape
        Tout = T if causal else T + 2 * pad_t - KW + 1
        Hout = Hin + 2 * pad_h - KH + 1
        pointwise = KH == 1 and KW == 1 and pad_h == 0 and pad_t == 0
        xp = None
        # first layer (Cin < 8, e.g. the loader's (B, 3, 40, T) float32 minibatch, no input gradient wanted): bring the input
        # to (T, B, H, 8) bf16 with zero channels and run the implicit GEMM with K padded by one empty tap to a multiple of 32
        if not pointwise and Ci < 8 and not ctx.needs_input_grad[0] and w16p is not None:
            xpad = _ops.pack_input_pad(x, (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), T, B, Hin, Ci, 8)
            y = _ops.conv_nt(xpad, w16p, b.detach() if b is not None else None, F32 if out_f32 else BF16, KH, KW, pad_h, pad_t,
                             +1, Tout, Hout)
            ctx.save_for_backward(xpad, w16t, wbwd)        # (T, B, H, 8) bf16: the operand of the implicit weight gradient
            ctx.params = (W, b)
            ctx.meta = (B, Ci, Hin, T, Co, KH, KW, pad_h, pad_t, Tout, Hout, False, x.dtype, False, 2)
            if out_f32:
                ctx._asr_mailbox = _GradMailbox(y, b)
            ctx._asr_biasbox = _BiasBox(y, b)
            return y.reshape(Tout, B, Hout, Co)
        # implicit GEMM (asr_conv_nt): no column matrix when the input is already physical bf16 and every 16-B chunk of a
        # virtual im2col row stays inside one tap
        xphys = x.permute(3, 0, 2, 1)
        if (not pointwise and xphys.dtype == BF16 and xphys.is_contiguous() and _ops.conv_implicit_ok(Ci, KH, KW)
                and w16.shape[1] == KH * KW * Ci):
            y = _ops.conv_nt(xphys, w16, b.detach() if b is not None else None, F32 if out_f32 else BF16, KH, KW, pad_h, pad_t,
                             +1, Tout, Hout)
            ctx.save_for_backward(xphys, w16t, wbwd)
            ctx.params = (W, b)
            ctx.meta = (B, Ci, Hin, T, Co, KH, KW, pad_h, pad_t, Tout, Hout, False, x.dtype, ctx.needs_input_grad[0], True)
            if out_f32:
                ctx._asr_mailbox = _GradMailbox(y, b)
            ctx._asr_biasbox = _BiasBox(y, b)
            return y.reshape(Tout, B, Hout, Co)
        if pointwise:
            xp = x.permute(3, 0, 2, 1)
            if not (xp.dtype == BF16 and xp.is_contiguous()):
                xp = _ops.permute4(x, (T, B, Hin, Ci), (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), BF16)
            col = xp.reshape(T * B * Hin, Ci)
            if Ci % 8:
                pointwise = False
        if not pointwise:
            col = _ops.im2col(x, (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), T, B, Hin, Ci, KH, KW, pad_h,
                              pad_t, Tout)
        y = _ops.gemm_nt(col, w16, b.detach() if b is not None else None, F32 if out_f32 else BF16)
        ctx.save_for_backward(col, w16t, wbwd)
        ctx.params = (W, b)
        ctx.meta = (B, Ci, Hin, T, Co, KH, KW, pad_h, pad_t, Tout, Hout, pointwise, x.dtype, ctx.needs_input_grad[0], False)
        if out_f32:
            ctx._asr_mailbox = _GradMailbox(y)
        ctx._asr_biasbox = _BiasBox(y, b)
        return y.reshape(Tout, B, Hout, Co)

    @staticmethod
    def backward(ctx, gy):
        col, w16t, wbwd = ctx.saved_tensors
        W, b = ctx.params
        B, Ci, Hin, T, Co, KH, KW, pad_h, pad_t, Tout, Hout, pointwise, xdtype, need_dx, implicit = ctx.meta
        bias_done = _take_bias_done(ctx)
        gy, gy_auto = _incoming_bf16(ctx, gy, Co)
        g2 = gy.reshape(Tout * B * Hout, Co)
        Kreal = KH * KW * Ci
        xphys = col if implicit else None        # the implicit forward saved the input, not a column matrix
        Kp = (Kreal + 7) // 8 * 8 if implicit else col.shape[1]
        w_is_param = isinstance(W, torch.nn.Parameter)
        if w_is_param:
            gW = grad_buffer(W)
        else:       # a derived weight (weight normalisation): hand its gradient back to the tape
            gW = torch.empty(W.shape, dtype=F32, device=gy.device)
            _ops.fill_(gW, 0.0)
        gb = grad_buffer(b) if b is not None else None

        def weight_grads():
            if implicit:            # no column matrix here either: the virtual im2col rows are gathered by the TN kernel
                Cs = xphys.shape[3]         # 8 for the first layer's zero-padded channels
                copies = _ops.conv_tn_copies(Co, Cs, KH, KW)       # one copy per XCD where a single tile takes hundreds of K splits
                scratch = torch.empty((copies, Co, KH * KW * Cs) if copies > 1 else (Co, KH * KW * Cs), dtype=F32, device=gy.device)
                _ops.fill_(scratch, 0.0)
                _ops.conv_tn_acc(g2, xphys, scratch, KH, KW, pad_h, pad_t, Tout, Hout)
                _ops.conv_weight_grad_unpack(scratch, gW, Cs)
            elif Kp == Kreal and KH == 1 and KW == 1:
                _ops.gemm_tn_acc(g2, col, gW.reshape(Co, Kreal))
            else:
                # weights are stored (Co, Ci, kh, kw); the GEMM produces (Co, (kh, kw, ci)): accumulate through a scratch
                scratch = torch.empty((Co, Kp), dtype=F32, device=gy.device)
                _ops.fill_(scratch, 0.0)
                _ops.gemm_tn_acc(g2, col, scratch)
                _ops.conv_weight_grad_unpack(scratch, gW)
            if gb is not None:
                if not bias_done:
                    _ops.colsum_acc(g2, gb)
                elif gy_auto is not None and getattr(ctx, "_asr_mailbox", None) is not None:
                    _ops.colsum_acc(gy_auto, gb)        # the part another consumer of y sent through autograd

        def input_grad():
            if not need_dx:
                return None
            if (not pointwise and KH == Hin and Hout == 1 and KW == 1 and pad_h == 0 and pad_t == 0 and Tout == T
                    and w16t.shape[0] == Kreal):
                # the recipes' "dense" convolution: a kernel as high as its input and one frame wide (run/ctc/cnn/model.py: ksize
                # (kernel_height, 1)) -- every input element belongs to exactly one output position, so dx is ONE plain product
                # gy . W^T whose columns (kh, ci) are the input's (height, channel) order.  As an implicit backward-data convolution
                # it summed KH taps per output row of which all but one lie outside the one-row gradient: 13 x the flops
                # (0.99 ms instead of 0.11 for the 13 x 128 -> 640 layer at T=1000, B=32)
                gp = _ops.gemm_nt(g2, w16t, None, BF16).reshape(T, B, Hin, Ci)
            elif wbwd is not None and not pointwise:      # implicit backward-data: no dcol matrix, no col2im pass
                gp = _ops.conv_nt(g2.reshape(Tout, B, Hout, Co), wbwd, None, BF16, KH, KW, pad_h, pad_t, -1, T, Hin).reshape(T, B, Hin, Ci)
            else:
                dcol = _ops.gemm_nt(g2, w16t, None, BF16)
                gp = dcol.reshape(T, B, Hin, Ci) if pointwise else _ops.col2im(dcol, T, B, Hin, Ci, KH, KW, pad_h, pad_t, Tout)
            if xdtype == F32:
                return _ops.bf16_to_f32(gp).permute(1, 3, 2, 0)
            return gp.permute(1, 3, 2, 0)

        if w_is_param:      # the side stream forks BEFORE the input-gradient product is queued: beside it, not behind it
            with _OnSide(g2, col, gy_auto):     # (gy_auto: read by the bias column sum of a second autograd consumer, as in _Dense)
                weight_grads()
            gx = input_grad()
            grads_queued(W, b)
        else:               # (derived weight: its gradient goes back to the tape on this stream -- the activation gradient first)
            gx = input_grad()
            weight_grads()
            grads_queued(b)
        return gx, (None if w_is_param else gW), None, None, None, None, None, None, None, None, None


def conv_weight_matrix(W):
    """(Co, Ci, kh, kw) f32 -> bf16 (Co, Kp) with k = (kh, kw, ci), zero padded to a multiple of 8."""
    return _ops.conv_weight_pack(W.contiguous())


def conv_weight_matrix_t(W):
    """bf16 (Kp, Co): the transposed matrix used by the backward-data GEMM."""
    return _ops.conv_weight_pack(W.contiguous(), transpose=True)


def conv_weight_matrix_bwd(W):
    """bf16 (Ci, kh*kw*Co): the operand of the implicit backward-data convolution."""
    return _ops.conv_weight_pack_bwd(W.contiguous())


def conv_weight_matrix_pad8(W):
    """first-layer operand of the implicit convolution: channels zero-padded to 8, k = (kh, kw, c8), rows padded with empty
    (zero) taps to a multiple of 32"""
    Co, Ci, KH, KW = W.shape
    Wp = torch.zeros((Co, 8, KH, KW), dtype=W.dtype, device=W.device)
    Wp[:, :Ci] = W
    return _ops.conv_weight_pack(Wp, Kp=(KH * KW * 8 + 31) // 32 * 32)


def convolution_2d(x, W, b, link, pad=(0, 0), causal=False, out_f32=False):
    """Cross-correlation over (height, time), stride 1 (asr/nn/nn.py:235-238 forces stride=1)."""
    pad_h, pad_t = (pad, pad) if isinstance(pad, int) else pad
    w16 = link.compute_copy("w16", W, conv_weight_matrix)
    w16t = link.compute_copy("w16t", W, conv_weight_matrix_t)
    wbwd = link.compute_copy("wbwd", W, conv_weight_matrix_bwd) if _ops.conv_implicit_ok(W.shape[0], W.shape[2], W.shape[3]) else None
    if x.dtype not in (F32, BF16):
        raise TypeError("convolution input must be float32 or bfloat16")
    w16p = link.compute_copy("w16p", W, conv_weight_matrix_pad8) if W.shape[1] < 8 else None
    y = _Conv2D.apply(x, W, b, w16, w16t, wbwd, w16p, int(pad_h), int(pad_t), bool(causal), bool(out_f32))
    return logical4(y)


def convolution_2d_given_weight(x, W, b, link, pad=(0, 0), causal=False, out_f32=False):
    """the same convolution with a weight that is itself a function of parameters (weight normalisation): the bf16
    matrices are rebuilt on every call"""
    pad_h, pad_t = (pad, pad) if isinstance(pad, int) else pad
    with torch.no_grad():
        w16 = conv_weight_matrix(W.detach())
        w16t = conv_weight_matrix_t(W.detach())
        wbwd = conv_weight_matrix_bwd(W.detach()) if _ops.conv_implicit_ok(W.shape[0], W.shape[2], W.shape[3]) else None
        w16p = conv_weight_matrix_pad8(W.detach()) if W.shape[1] < 8 else None
    y = _Conv2D.apply(x, W, b, w16, w16t, wbwd, w16p, int(pad_h), int(pad_t), bool(causal), bool(out_f32))
    return logical4(y)


# ---------------------------------------------------------------------------------------------- first block in one pass
CONV_MP = [_lib_debug("conv_mp", 1) != 0]        # ASR_DEBUG conv_mp=0: convolution, maxout and pooling as separate passes (comparison)


def conv_weight_matrix_pad8_128(W):
    """operand of the fused first block (csrc/conv_first.hip): conv_weight_matrix_pad8 with rows of exactly 128 entries (16 taps of 8)"""
    Co, Ci, KH, KW = W.shape
    Wp = torch.zeros((Co, 8, KH, KW), dtype=W.dtype, device=W.device)
    Wp[:, :Ci] = W
    return _ops.conv_weight_pack(Wp, Kp=128)


class _ConvMaxoutPool(torch.autograd.Function):
    """x (B, Ci < 8, H, T), no gradient wanted (the loader's minibatch); W (Co, Ci, kh, kw) f32 master or derived; b (Co) or None.
    Output physical (Tout, B, Hp, Co / 2): Maxout(2) and MaxPooling2D((k, 1)) of the convolution, which is never written."""

    @staticmethod
    def forward(ctx, x, W, b, w16q, pad_h, pad_t, causal, k):
        B, Ci, Hin, T = x.shape
        Co, _, KH, KW = W.shape
        Tout = T if causal else T + 2 * pad_t - KW + 1
        Hout = Hin + 2 * pad_h - KH + 1
        xpad = _ops.pack_input_pad(x, (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), T, B, Hin, Ci, 8)
        y, idx = _ops.conv_mp_fwd(xpad, w16q, b.detach() if b is not None else None, KH, KW, pad_h, pad_t, Tout, Hout, k)
        ctx.save_for_backward(xpad, idx)
        ctx.params = (W, b)
        ctx.meta = (KH, KW, pad_h, pad_t, Hout, k)
        return y

    @staticmethod
    def backward(ctx, gy):
        xpad, idx = ctx.saved_tensors
        W, b = ctx.params
        KH, KW, pad_h, pad_t, Hout, k = ctx.meta
        gy = gy.contiguous() if gy.dtype == BF16 else _ops.cast_bf16(gy.contiguous())
        w_is_param = isinstance(W, torch.nn.Parameter)
        if w_is_param:
            gW = grad_buffer(W)
        else:       # a derived weight (weight normalisation): hand its gradient back to the tape
            gW = torch.empty(W.shape, dtype=F32, device=gy.device)
            _ops.fill_(gW, 0.0)
        gb = grad_buffer(b) if b is not None else None
        if w_is_param:
            with _OnSide(gy, xpad, idx):
                _ops.conv_mp_bwd(gy, idx, xpad, gW, gb, KH, KW, pad_h, pad_t, Hout, k)
            grads_queued(W, b)
        else:
            _ops.conv_mp_bwd(gy, idx, xpad, gW, gb, KH, KW, pad_h, pad_t, Hout, k)
            grads_queued(b)
        return None, (None if w_is_param else gW), None, None, None, None, None, None


def convolution_maxout_pool_ok(x, wshape, pad, causal, k):
    pad_h, pad_t = (pad, pad) if isinstance(pad, int) else pad
    if not CONV_MP[0] or x.dim() != 4 or x.requires_grad or x.dtype not in (F32, BF16) or len(wshape) != 4 or wshape[1] != x.shape[1]:
        return False
    Co, Ci, KH, KW = wshape
    if not _ops.conv_mp_ok(Ci, KH, KW, Co, k) or Ci >= 8:
        return False
    return x.shape[2] + 2 * pad_h - KH + 1 >= 1 and (x.shape[3] if causal else x.shape[3] + 2 * pad_t - KW + 1) >= 1


def convolution_maxout_pool(x, W, b, link, pad, causal, k, given_weight=False):
    """convolution_2d -> maxout(., 2) -> max_pooling_2d(., (k, 1)) of the FIRST layer (fewer than 8 input channels, no input gradient) as
    one pass forward and one backward; None where the fused kernels do not serve the layer (the caller then runs the three functions)."""
    pad_h, pad_t = (pad, pad) if isinstance(pad, int) else pad
    if not convolution_maxout_pool_ok(x, tuple(W.shape), pad, causal, k):
        return None
    if given_weight:
        with torch.no_grad():
            w16q = conv_weight_matrix_pad8_128(W.detach())
    else:
        w16q = link.compute_copy("w16q", W, conv_weight_matrix_pad8_128)
    return logical4(_ConvMaxoutPool.apply(x, W, b, w16q, int(pad_h), int(pad_t), bool(causal), int(k)))


# ---------------------------------------------------------------------------------------------- dense (1x1 over time)
class _Dense(torch.autograd.Function):
    """rows (T*B, Din) bf16 @ W (Dout, Din)^T + b."""

    @staticmethod
    def forward(ctx, x2, W, b, w16, w16t, out_f32):
        y = _ops.gemm_nt(x2, w16, b.detach() if b is not None else None, F32 if out_f32 else BF16)
        ctx.save_for_backward(x2, w16t)
        ctx.params = (W, b)
        ctx.need_dx = ctx.needs_input_grad[0]
        if out_f32:
            ctx._asr_mailbox = _GradMailbox(y, b)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, w16t = ctx.saved_tensors
        W, b = ctx.params
        bias_done = _take_bias_done(ctx)
        gy, gy_auto = _incoming_bf16(ctx, gy, W.shape[0])
        gW = grad_buffer(W).reshape(W.shape[0], -1)
        gb = grad_buffer(b) if b is not None else None
        with _OnSide(gy, x2, gy_auto):      # forked BEFORE the input-gradient product is queued: beside it, not behind it
            _ops.gemm_tn_acc(gy, x2, gW)
            if gb is not None:
                if not bias_done:
                    _ops.colsum_acc(gy, gb)
                elif gy_auto is not None:
                    _ops.colsum_acc(gy_auto, gb)        # the part another consumer of y sent through autograd
        gx = _ops.gemm_nt(gy, w16t, None, BF16) if ctx.need_dx else None
        grads_queued(W, b)
        return gx, None, None, None, None, None


def dense(x2, W, b, link, out_f32=False):
    w16 = link.compute_copy("w16", W, lambda w: _ops.cast_bf16(w.reshape(w.shape[0], -1)), "plain")
    w16t = link.compute_copy("w16t", W, lambda w: _ops.cast_bf16(w.reshape(w.shape[0], -1), transpose=True), "t_first")
    return _Dense.apply(x2, W, b, w16, w16t, bool(out_f32))


def convolution_1d(x, W, b, link, out_f32=False):
    """ConvolutionND with ksize 1 over (B, C, T) (asr/nn/convolution_1d.py:7-38): a per-frame affine map."""
    p = phys3(x)
    T, B, D = p.shape
    y = dense(p.reshape(T * B, D), W, b, link, out_f32)
    return logical3(y.reshape(T, B, -1))


def linear(x, W, b, link):
    """chainer.links.Linear on (N, D)."""
    p, _ = _phys_any(x)
    return dense(p, W, b, link)


# ---------------------------------------------------------------------------------------------- activations / pooling
class _Maxout2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        ctx.save_for_backward(p)
        return _ops.maxout2_fwd(p)

    @staticmethod
    def backward(ctx, gy):
        (p,) = ctx.saved_tensors
        return _ops.maxout2_bwd(p, gy.contiguous())


def maxout(x, pool_size=2, axis=1):
    """chainer.functions.maxout(x, pool_size, axis=1): max over groups of adjacent channels (asr/nn/nn.py:45-50)."""
    if pool_size != 2 or axis != 1:
        raise NotImplementedError("only maxout(x, 2, axis=1) is on the HIP path (the reference uses nothing else)")
    p, back = _phys_any(x)
    return back(_Maxout2.apply(p))


class _MaxPoolH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, k):
        ctx.save_for_backward(p)
        ctx.k = k
        return _ops.maxpool_h_fwd(p, k)

    @staticmethod
    def backward(ctx, gy):
        (p,) = ctx.saved_tensors
        return _ops.maxpool_h_bwd(p, gy.contiguous(), ctx.k), None


BIAS_FROM_POOL_MIN_NUMEL = [1 << 26]


class _Maxout2PoolH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, k, sole):
        ctx.save_for_backward(p)
        ctx.k = k
        # the only consumer of a convolution's output: its backward pass forms that output's whole gradient, so the bias gradient
        # (the column sums) comes out of the same pass
        # (worth it on a large output only: 141 us against 128 + 146 us for the first block of the BASELINE model, but 73 against
        # 32 + 32 us for the second -- the sums end in one atomic per channel and workgroup on one address each)
        box = _producer_box(p, "_asr_biasbox", 12, BF16) if (sole and p.numel() >= BIAS_FROM_POOL_MIN_NUMEL[0]) else None
        if box is not None and not (isinstance(box.bias, torch.nn.Parameter) and box.bias.numel() == p.shape[3]
                                    and _ops.maxout2_pool_bwd_db_ok(p.shape[3] // 2)):
            box = None
        ctx.biasbox = box
        return _ops.maxout2_pool_fwd(p, k)

    @staticmethod
    def backward(ctx, gy):
        (p,) = ctx.saved_tensors
        db = None
        if ctx.biasbox is not None and not ctx.biasbox.bias_done:
            db = grad_buffer(ctx.biasbox.bias).reshape(-1)
            ctx.biasbox.bias_done = True
        return _ops.maxout2_pool_bwd(p, gy.contiguous(), ctx.k, db), None, None


def maxout_max_pooling(x, k, sole_consumer=False):
    """maxout(x, 2) followed by max_pooling_2d(., (k, 1)) -- the tail of every conv block of the recipes -- as one pass
    over the convolution output (asr.nn containers use it when the two layers follow each other); falls back to the two
    functions when the layout does not allow the fused kernels."""
    p = phys4(x)
    if _ops.maxout2_pool_ok(p):
        return logical4(_Maxout2PoolH.apply(p, int(k), bool(sole_consumer)))
    return max_pooling_2d(maxout(x, 2), (k, 1))


def max_pooling_2d(x, ksize, stride=None, pad=0, cover_all=True):
    """chainer.functions.max_pooling_2d restricted to what the reference uses: ksize (k, 1), stride = ksize,
    pad 0, cover_all True (asr/nn/nn.py:95-103 passes nothing else)."""
    kh, kw = (ksize, ksize) if isinstance(ksize, int) else ksize
    sh, sw = (kh, kw) if stride is None else ((stride, stride) if isinstance(stride, int) else stride)
    if kw != 1 or sw != 1 or sh != kh or pad not in (0, (0, 0)) or not cover_all:
        raise NotImplementedError("max_pooling_2d on the HIP path: ksize (k, 1), stride (k, 1), pad 0, cover_all")
    return logical4(_MaxPoolH.apply(phys4(x), int(kh)))


class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, kind, alpha):
        ctx.save_for_backward(p)
        ctx.meta = (kind, alpha)
        return _ops.activation_fwd(p, kind, alpha)

    @staticmethod
    def backward(ctx, gy):
        (p,) = ctx.saved_tensors
        kind, alpha = ctx.meta
        return _ops.activation_bwd(p, gy.contiguous(), kind, alpha), None, None


def _act(x, kind, alpha=0.0):
    p, back = _phys_any(x)
    return back(_Activation.apply(p, kind, float(alpha)))


def relu(x):
    return _act(x, "relu")


def clipped_relu(x, z=20.0):
    return _act(x, "clipped_relu", z)


def leaky_relu(x, slope=0.2):
    return _act(x, "leaky_relu", slope)


def elu(x, alpha=1.0):
    return _act(x, "elu", alpha)


def sigmoid(x):
    return _act(x, "sigmoid")


def tanh(x):
    return _act(x, "tanh")


def hard_sigmoid(x):
    return _act(x, "hard_sigmoid")


def softplus(x, beta=1.0):
    return _act(x, "softplus", beta)


class _GLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        ctx.save_for_backward(p)
        return _ops.glu_fwd(p)

    @staticmethod
    def backward(ctx, gy):
        (p,) = ctx.saved_tensors
        return _ops.glu_bwd(p, gy.contiguous())


def glu(x):
    """A, B = split_axis(x, 2, axis=1); A * sigmoid(B)  (asr/nn/nn.py:279-280)."""
    p, back = _phys_any(x)
    return back(_GLU.apply(p))


class _Dropout(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, ratio, seed):
        ctx.meta = (ratio, seed)
        return _ops.dropout(p, ratio, seed)

    @staticmethod
    def backward(ctx, gy):
        ratio, seed = ctx.meta
        return _ops.dropout(gy.contiguous(), ratio, seed), None, None


_dropout_counter = [0]
train_mode = [True]          # chainer.config.train


def dropout(x, ratio=0.5):
    if ratio == 0 or not train_mode[0]:
        return x
    p, back = _phys_any(x)
    _dropout_counter[0] += 1
    seed = (torch.initial_seed() * 1000003 + _dropout_counter[0]) & 0xffffffff
    return back(_Dropout.apply(p, float(ratio), seed))


class _Add(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return _ops.add_bf16(a, b)

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    """residual connection ``y += x`` (asr/nn/nn.py:322-328)."""
    pa, back = _phys_any(a)
    pb, _ = _phys_any(b)
    return back(_Add.apply(pa, pb))


# ---------------------------------------------------------------------------------------------- the rest of asr.nn's function layers
def _channel_rows(x, axis):
    """physical tensor whose LAST axis is the logical axis 1 (the channels): the only axis these functions are offered on"""
    if axis not in (1, 1 - x.dim()):
        raise NotImplementedError("on the HIP path this function works along axis 1 (the channels), as asr.nn uses it")
    return _phys_any(x)


class _CReLU(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p):
        ctx.save_for_backward(p)
        return _ops.crelu_fwd(p)

    @staticmethod
    def backward(ctx, gy):
        (p,) = ctx.saved_tensors
        return _ops.crelu_bwd(p, gy.contiguous())


def crelu(x, axis=1):
    """chainer.functions.crelu (asr/nn/nn.py:18-23): concat(relu(x), relu(-x)) along axis 1"""
    p, back = _channel_rows(x, axis)
    return back(_CReLU.apply(p))


class _Softmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, log_form):
        y = _ops.softmax_fwd(p, log_form)
        ctx.save_for_backward(y)
        ctx.log_form = log_form
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return _ops.softmax_bwd(y, gy.contiguous(), ctx.log_form), None


def softmax(x, axis=1):
    """chainer.functions.softmax (asr/nn/nn.py:58-63)"""
    p, back = _channel_rows(x, axis)
    return back(_Softmax.apply(p, False))


def log_softmax(x, axis=1):
    """chainer.functions.log_softmax (asr/nn/nn.py:42-43)"""
    p, back = _channel_rows(x, axis)
    return back(_Softmax.apply(p, True))


def _height_pooling_args(ksize, stride, pad, what):
    kh, kw = (ksize, ksize) if isinstance(ksize, int) else ksize
    sh, sw = (kh, kw) if stride is None else ((stride, stride) if isinstance(stride, int) else stride)
    if kw != 1 or sw != 1 or sh != kh or pad not in (0, (0, 0)):
        raise NotImplementedError("%s on the HIP path: ksize (k, 1), stride (k, 1), pad 0 (pooling over the mel axis, as the recipes pool)" % what)
    return int(kh)


class _AvgPoolH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, k):
        ctx.meta = (p.shape[2], k)
        return _ops.avgpool_h_fwd(p, k)

    @staticmethod
    def backward(ctx, gy):
        H, k = ctx.meta
        return _ops.avgpool_h_bwd(gy.contiguous(), H, k), None


def average_pooling_2d(x, ksize, stride=None, pad=0):
    """chainer.functions.average_pooling_2d (asr/nn/nn.py:77-84): whole windows only (Chainer's average pooling has no cover_all)"""
    k = _height_pooling_args(ksize, stride, pad, "average_pooling_2d")
    return logical4(_AvgPoolH.apply(phys4(x), k))


def average_pooling_nd(x, ksize, stride=None, pad=0):
    """chainer.functions.average_pooling_nd on a 4-d array = the 2-d one (asr/nn/nn.py:86-93)"""
    if x.dim() != 4:
        raise NotImplementedError("average_pooling_nd on the HIP path: 4-d (B, C, H, T) arrays")
    return average_pooling_2d(x, ksize, stride, pad)


def max_pooling_nd(x, ksize, stride=None, pad=0, cover_all=True):
    """chainer.functions.max_pooling_nd on a 4-d array = the 2-d one (asr/nn/nn.py:105-113)"""
    if x.dim() != 4:
        raise NotImplementedError("max_pooling_nd on the HIP path: 4-d (B, C, H, T) arrays")
    return max_pooling_2d(x, ksize, stride, pad, cover_all)


class _UnpoolH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, k, Hout):
        ctx.meta = (p.shape[2], k)
        return _ops.unpool_h_fwd(p, k, Hout)

    @staticmethod
    def backward(ctx, gy):
        H, k = ctx.meta
        return _ops.unpool_h_bwd(gy.contiguous(), H, k), None, None


def unpooling_2d(x, ksize, stride=None, pad=0, outsize=None, cover_all=True):
    """chainer.functions.unpooling_2d (asr/nn/nn.py:123-133) for ksize (k, 1), stride = ksize: every row repeated over its window;
    output height k (H - 1) + 1 with cover_all, k H without (chainer.utils.conv.get_deconv_outsize), or `outsize`"""
    k = _height_pooling_args(ksize, stride, pad, "unpooling_2d")
    p = phys4(x)
    H = p.shape[2]
    Hout = (k * (H - 1) + 1 if cover_all else k * H) if outsize is None else int(outsize[0] if isinstance(outsize, (tuple, list)) else outsize)
    if not (k * (H - 1) < Hout <= k * H):
        raise ValueError("outsize %d does not fit %d input rows with ksize %d" % (Hout, H, k))
    return logical4(_UnpoolH.apply(p, k, Hout))


class _UpsampleH(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, idx, k, Hout):
        ctx.save_for_backward(idx)
        ctx.k = k
        return _ops.upsample_h_fwd(p, idx, k, Hout)

    @staticmethod
    def backward(ctx, gy):
        (idx,) = ctx.saved_tensors
        return _ops.upsample_h_bwd(gy.contiguous(), idx, ctx.k), None, None, None


def max_pooling_2d_indexes(x, ksize):
    """the `indexes` attribute of Chainer's MaxPooling2D function object for max_pooling_2d(x, (k, 1)): where inside its window every
    maximum sits (first one on ties) -- what upsampling_2d takes.  Logical (B, C, Hout, T) uint8."""
    kh, kw = (ksize, ksize) if isinstance(ksize, int) else ksize
    if kw != 1:
        raise NotImplementedError("max pooling on the HIP path: ksize (k, 1)")
    idx = _ops.maxpool_h_indexes(phys4(x.detach()), int(kh))            # physical (T, B, Hout, C)
    return idx.permute(1, 3, 2, 0)


def upsampling_2d(x, indexes, ksize, stride=None, pad=0, outsize=None, cover_all=True):
    """chainer.functions.upsampling_2d (asr/nn/nn.py:135-146) for ksize (k, 1), stride = ksize: x[b][c][h][t] goes to row h k + indexes of
    the output, every other entry is zero; output height as unpooling_2d.  `indexes`: logical (B, C, H, T) integers in [0, k) -- see
    max_pooling_2d_indexes."""
    k = _height_pooling_args(ksize, stride, pad, "upsampling_2d")
    p = phys4(x)
    H = p.shape[2]
    Hout = (k * (H - 1) + 1 if cover_all else k * H) if outsize is None else int(outsize[0] if isinstance(outsize, (tuple, list)) else outsize)
    if not (k * (H - 1) < Hout <= k * H):
        raise ValueError("outsize %d does not fit %d input rows with ksize %d" % (Hout, H, k))
    if tuple(indexes.shape) != tuple(x.shape):
        raise ValueError("indexes %s must have the shape of x %s" % (tuple(indexes.shape), tuple(x.shape)))
    idx = indexes.permute(3, 0, 2, 1)
    if idx.dtype != torch.uint8 or not idx.is_contiguous():
        idx = idx.to(torch.uint8).contiguous()
    return logical4(_UpsampleH.apply(p, idx, k, Hout))


class _SppMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, height):
        y, pos = _ops.spp_fwd(p, height, ctx.needs_input_grad[0])
        ctx.pos, ctx.shape, ctx.height = pos, tuple(p.shape), height
        return y

    @staticmethod
    def backward(ctx, gy):
        return _ops.spp_bwd(gy.contiguous(), ctx.pos, ctx.shape, ctx.height), None


def spatial_pyramid_pooling_2d(x, pyramid_height, pooling_class=None):
    """chainer.functions.spatial_pyramid_pooling_2d (asr/nn/nn.py:115-121): x (B, C, H, T) -> (B, C (4^height - 1) / 3, 1, 1), for every
    level l < height the maxima over 2^l x 2^l bins of the (H, T) plane, [c][by][bx] per level, levels concatenated along axis 1.
    Chainer accepts max pooling only; `pooling_class` may be None, "max", nn.MaxPooling2D or anything whose name contains MaxPooling."""
    name = pooling_class if isinstance(pooling_class, str) else getattr(pooling_class, "__name__", type(pooling_class).__name__ if pooling_class is not None else "max")
    if pooling_class is not None and "max" not in name.lower():
        raise NotImplementedError("spatial_pyramid_pooling_2d: max pooling only (as in Chainer)")
    p = phys4(x)
    B, C = x.shape[0], x.shape[1]
    y = _SppMax.apply(p, int(pyramid_height))              # (B, bins, C), bins level after level
    parts, o = [], 0
    for l in range(int(pyramid_height)):
        n = 4 ** l
        parts.append(y[:, o:o + n, :].permute(0, 2, 1).reshape(B, C * n))        # [c][by][bx]
        o += n
    return torch.cat(parts, dim=1).reshape(B, -1, 1, 1)


class _GaussianNoise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, p, std, seed):
        return _ops.gaussian_noise(p, std, seed)

    @staticmethod
    def backward(ctx, gy):
        return gy, None, None


def gaussian_noise(x, std):
    """x + N(0, std^2) in train mode (asr/nn/nn.py:220-231: ln_var = log(std^2), mean 0 -- the reference's `mean` is unused)"""
    if not train_mode[0]:
        return x
    p, back = _phys_any(x)
    _dropout_counter[0] += 1
    seed = (torch.initial_seed() * 1000003 + 7919 * _dropout_counter[0]) & 0xffffffff
    return back(_GaussianNoise.apply(p, float(std), seed))


# ---------------------------------------------------------------------------------------------- layer normalisation
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, gamma, beta, C, out_f32, mailbox, tb):
        # float32 rows normalised over their whole width = logits normalised over the vocabulary of each (t, b) frame: a
        # CTC-family loss on them may leave a recipe instead of a gradient (see _CtcBox), and finds the log-sum-exp of every row
        # already there (formed by the normalisation while it had the row in registers)
        per_frame = out_f32 and tb is not None and x2.dtype == F32 and C == x2.shape[1] and C % 4 == 0 and C <= 4096
        want_lse = per_frame and FUSE_CTC_INTO_LAYERNORM[0] and x2.data_ptr() % 16 == 0
        res = _ops.layernorm_fwd(x2, gamma.detach(), beta.detach(), C, F32 if out_f32 else BF16, want_lse)
        y, mean, rstd = res[:3]
        ctx.save_for_backward(x2, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.meta = (C, ctx.needs_input_grad[0], mailbox)
        if per_frame:
            ctx._asr_ctc_box = _CtcBox(y, tb[0], tb[1], C, res[3] if want_lse else None)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        C, need_dx, mailbox = ctx.meta
        handover = mailbox is not None and need_dx      # the producer of x2 takes its gradient in bf16
        dx_dtype = BF16 if handover else x2.dtype
        box = getattr(ctx, "_asr_ctc_box", None)
        recipes = box.take() if box is not None else []
        dx = None
        if recipes:
            LAST_FUSED_RECIPES[0] = len(recipes)
            T, B, _ = box.shape
            # the whole gradient of x2 is formed in this sweep: its column sums are the bias gradient of the projection in front
            dxsum = None
            if handover and _is_zero_token(gy) and mailbox.bias is not None and isinstance(mailbox.bias, torch.nn.Parameter) \
                    and mailbox.bias.numel() == C:
                dxsum = grad_buffer(mailbox.bias).reshape(-1)
                mailbox.bias_done = True
            dx = _ops.layernorm_ctc_bwd(x2, gamma.detach(), beta.detach(), mean, rstd, T, B, dx_dtype, grad_buffer(gamma),
                                        grad_buffer(beta), need_dx, recipes, dxsum)
        if not recipes or not _is_zero_token(gy):       # a gradient that did arrive through autograd
            dx2 = _ops.layernorm_bwd(x2, gy.contiguous(), gamma.detach(), mean, rstd, C, dx_dtype, grad_buffer(gamma),
                                     grad_buffer(beta), need_dx)
            if dx is None:
                dx = dx2
            elif need_dx:
                dx = _ops.add_bf16(dx, dx2) if dx_dtype == BF16 else dx.add_(dx2)
        grads_queued(gamma, beta)
        if handover:
            mailbox.post(dx)
            dx = _zero_token(x2.shape, x2.device)
        return dx, None, None, None, None, None, None


def layer_normalization(x, gamma, beta, out_f32=False):
    """normalize_layer over axes (1, 2) + scale/bias on axis 1 (asr/nn/nn.py:260-265, asr/nn/layernorm.py:29-64)."""
    if x.dim() == 4:
        p = x.permute(3, 0, 2, 1)
        if not (p.is_contiguous() and p.dtype in (BF16, F32)):
            p = phys4(x)
        T, B, H, C = p.shape
        x2 = p.reshape(T * B, H * C)
        y = _LayerNorm.apply(x2, gamma, beta, C, bool(out_f32), _producer_mailbox(x2), (T, B) if H == 1 else None)
        return logical4(y.reshape(T, B, H, C))
    if x.dim() == 3:
        # (B, V, T): the reference normalises over V AND T jointly (axes 1, 2).  Physical rows are (t, b); joint
        # statistics over time need the (B, T*V) arrangement: one row per utterance, channel = index % V.
        Bn, V, T = x.shape
        rows = x.permute(0, 2, 1)                # (B, T, V)
        if not (rows.is_contiguous() and rows.dtype in (BF16, F32)):
            rows = _ToPhys.apply(x, (0, 2, 1))
        y = _LayerNorm.apply(rows.reshape(Bn, T * V), gamma, beta, V, bool(out_f32), None, None)
        return y.reshape(Bn, T, V).permute(0, 2, 1)
    raise ValueError("layer normalisation expects a 3-d or 4-d input")


class _BatchNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2, gamma, beta, avg_mean, avg_var, eps, decay, train):
        if train:
            mean, rstd = _ops.batchnorm_stats(x2, eps, decay, avg_mean, avg_var)
        else:       # inference: the running statistics (chainer.config.train == False)
            mean, rstd = avg_mean, _rsqrt_eps(avg_var, eps)
        y = _ops.batchnorm_fwd(x2, mean, rstd, gamma.detach(), beta.detach())
        ctx.save_for_backward(x2, mean, rstd)
        ctx.params = (gamma, beta)
        ctx.meta = (ctx.needs_input_grad[0], train)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, mean, rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        need_dx, train = ctx.meta
        gy = gy.contiguous()
        if not train:
            raise RuntimeError("backward through BatchNormalization in test mode is not supported (fixed statistics)")
        dx = _ops.batchnorm_bwd(x2, gy, mean, rstd, gamma.detach(), grad_buffer(gamma), grad_buffer(beta), need_dx)
        grads_queued(gamma, beta)
        return dx, None, None, None, None, None, None, None


def _rsqrt_eps(var, eps):
    """1 / sqrt(var + eps) on C values (asr_rsqrt_eps): the inference form's rstd from the running variance"""
    return _ops.rsqrt_eps(var, eps)


def batch_normalization(x, gamma, beta, avg_mean, avg_var, eps=2e-5, decay=0.9):
    """chainer.links.BatchNormalization on a (B, C, H, T) [or (B, C, T), (B, C)] activation: per-channel statistics over
    every other axis; train mode (functions.train_mode) uses and records the batch statistics."""
    train = bool(train_mode[0])
    if x.dim() == 4:
        p = x.permute(3, 0, 2, 1)
        if not (p.is_contiguous() and p.dtype == BF16):
            p = phys4(x)
        T, B, H, C = p.shape
        y = _BatchNorm.apply(p.reshape(T * B * H, C), gamma, beta, avg_mean, avg_var, eps, decay, train)
        return logical4(y.reshape(T, B, H, C))
    if x.dim() == 3:
        p = x.permute(2, 0, 1)
        if not (p.is_contiguous() and p.dtype == BF16):
            p = phys3(x)
        T, B, C = p.shape
        y = _BatchNorm.apply(p.reshape(T * B, C), gamma, beta, avg_mean, avg_var, eps, decay, train)
        return logical3(y.reshape(T, B, C))
    raise ValueError("batch normalisation expects a 3-d or 4-d input")


# ---------------------------------------------------------------------------------------------- GRU
def _cast_transposed_per_direction(w):
    """(ndir, 3H, H) f32 -> (ndir, H, 3H) bf16, written in place (no torch.stack copy)."""
    out = torch.empty((w.shape[0], w.shape[2], w.shape[1]), dtype=BF16, device=w.device)
    for d in range(w.shape[0]):
        _ops.cast_bf16(w[d], transpose=True, out=out[d])
    return out


class _GRU(torch.autograd.Function):
    """x rows (T*B, I) bf16 -> y rows (T*B, H) bf16 (directions summed)."""

    @staticmethod
    def forward(ctx, x2, w_ih, w_hh, b_ih, b_hh, copies, T, B, H, ndir, x_len):
        wih16, wih16t, whh16, whh16t = copies
        gi = _ops.gemm_nt(x2, wih16, b_ih.detach().reshape(-1), _ops.gru_gi_dtype(T, B, H, ndir))
        _empty_chip_for_recurrence()
        y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, b_hh.detach().reshape(-1), T, B, H, ndir, x_len)
        ctx.save_for_backward(x2, hseq, hseq16, gates, wih16t, whh16t)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        ctx.meta = (T, B, H, ndir, ctx.needs_input_grad[0], x_len)
        return y

    @staticmethod
    def backward(ctx, gy):
        x2, hseq, hseq16, gates, wih16t, whh16t = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh = ctx.params
        T, B, H, ndir, need_dx, x_len = ctx.meta
        gy = gy.contiguous()
        _empty_chip_for_recurrence()
        dgi, dgh = _ops.gru_bwd(gy, gates, hseq, whh16t, T, B, H, ndir, grad_buffer(b_ih).reshape(-1),
                                grad_buffer(b_hh).reshape(-1), x_len)
        gwih = grad_buffer(w_ih).reshape(ndir * 3 * H, -1)
        gwhh = grad_buffer(w_hh).reshape(ndir, 3 * H, H)
        with _OnSide(dgi, dgh, x2, hseq16):
            products = [(dgi, x2, gwih)]
            if T > 1:               # dW_hh[d] = sum_t dgh_t (x) h_{t-1}: the two directions differ by the sign of the shift only
                for d in range(ndir):
                    a = dgh[:, d * 3 * H:(d + 1) * 3 * H]
                    h = hseq16[:, d * H:(d + 1) * H]
                    products.append((a[B:], h[:-B], gwhh[d]) if d == 0 else (a[:-B], h[B:], gwhh[d]))
            _ops.gemm_tn_acc_group(products)
        # queued after the side stream's fork, so that the weight gradients start beside it and not behind it
        gx = _ops.gemm_nt(dgi, wih16t, None, BF16) if need_dx else None        # the only product on the critical path
        grads_queued(w_ih, w_hh, b_ih, b_hh)
        return gx, None, None, None, None, None, None, None, None, None, None


class _GRUState(torch.autograd.Function):
    """_GRU with a given initial state: x rows (T*B, I) bf16, hx (ndir, B, H) float32 -> y rows (T*B, H) bf16 (directions summed) and
    the final state hy (ndir, B, H) float32 -- chainer.links.NStepGRU's hx / hy.  One launch per time step (asr_gru_fwd_state)."""

    @staticmethod
    def forward(ctx, x2, w_ih, w_hh, b_ih, b_hh, hx, copies, T, B, H, ndir, x_len):
        wih16, wih16t, whh16, whh16t = copies
        gi = _ops.gemm_nt(x2, wih16, b_ih.detach().reshape(-1), F32)
        hx = hx.detach().to(F32).contiguous()
        y, hseq, hseq16, gates = _ops.gru_fwd_state(gi, whh16, b_hh.detach().reshape(-1), hx, T, B, H, ndir, x_len)
        h4 = hseq.reshape(T, B, ndir, H)
        hy = torch.stack([h4[T - 1, :, 0]] + ([h4[0, :, 1]] if ndir == 2 else []), dim=0).contiguous()
        ctx.save_for_backward(x2, hseq, hseq16, gates, wih16t, whh16t, hx)
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        ctx.meta = (T, B, H, ndir, ctx.needs_input_grad[0], x_len)
        return y, hy

    @staticmethod
    def backward(ctx, gy, ghy):
        x2, hseq, hseq16, gates, wih16t, whh16t, hx = ctx.saved_tensors
        w_ih, w_hh, b_ih, b_hh = ctx.params
        T, B, H, ndir, need_dx, x_len = ctx.meta
        if gy is None:
            gy = torch.zeros((T * B, H), dtype=BF16, device=hseq.device)
        dhy = None if ghy is None else ghy.to(F32).contiguous()
        dgi, dgh, dhx = _ops.gru_bwd_state(gy.contiguous(), gates, hseq, hx, dhy, whh16t, T, B, H, ndir, grad_buffer(b_ih).reshape(-1),
                                           grad_buffer(b_hh).reshape(-1), x_len)
        gwih = grad_buffer(w_ih).reshape(ndir * 3 * H, -1)
        gwhh = grad_buffer(w_hh).reshape(ndir, 3 * H, H)
        hx16 = hx.to(BF16)
        products = [(dgi, x2, gwih)]
        for d in range(ndir):           # dW_hh[d] = sum_t dgh_t (x) h_{t-1}, h_{-1} = hx
            a = dgh[:, d * 3 * H:(d + 1) * 3 * H]
            h = hseq16[:, d * H:(d + 1) * H]
            if T > 1:
                products.append((a[B:], h[:-B], gwhh[d]) if d == 0 else (a[:-B], h[B:], gwhh[d]))
            products.append((a[:B] if d == 0 else a[(T - 1) * B:], hx16[d], gwhh[d]))
        for a, b, c in products:
            _ops.gemm_tn_acc(a, b, c)
        gx = _ops.gemm_nt(dgi, wih16t, None, BF16) if need_dx else None
        grads_queued(w_ih, w_hh, b_ih, b_hh)
        return gx, None, None, None, None, dhx, None, None, None, None, None, None


def gru(x, w_ih, w_hh, b_ih, b_hh, link, ndir, x_length=None, hx=None):
    """x logical (B, I, T); parameters stacked over directions: w_ih (ndir, 3H, I), w_hh (ndir, 3H, H),
    b_ih / b_hh (ndir, 3H).  Returns logical (B, H, T), the directions summed.
    x_length (B) int32 device tensor or None: frames per utterance -- the recurrences then run every utterance over its own
    length as chainer.links.NStepBiGRU does (asr/nn/nn.py:3): the reverse direction of utterance b starts at x_length[b] - 1,
    the output is zero beyond it and the padding receives / passes no gradient.
    hx (ndir, B, H) float32 or None: the initial state; given one, the result is (y, hy) with hy (ndir, B, H) the final state
    (NStepGRU's hx / hy; this form runs one launch per time step: _GRUState)."""
    p = phys3(x)
    T, B, I = p.shape
    H = w_hh.shape[2]
    copies = (
        link.compute_copy("wih16", w_ih, lambda w: _ops.cast_bf16(w.reshape(-1, w.shape[-1])), "plain"),
        link.compute_copy("wih16t", w_ih, lambda w: _ops.cast_bf16(w.reshape(-1, w.shape[-1]), transpose=True), "t_last"),
        link.compute_copy("whh16", w_hh, lambda w: _ops.cast_bf16(w.reshape(-1, w.shape[-1])).reshape(w.shape), "plain"),
        link.compute_copy("whh16t", w_hh, _cast_transposed_per_direction, "t_each"),
    )
    if hx is not None:
        if tuple(hx.shape) != (ndir, B, H):
            raise ValueError("hx must have shape (ndir, B, H) = %s, got %s" % ((ndir, B, H), tuple(hx.shape)))
        y, hy = _GRUState.apply(p.reshape(T * B, I), w_ih, w_hh, b_ih, b_hh, hx, copies, T, B, H, ndir, x_length)
        return logical3(y.reshape(T, B, H)), hy
    y = _GRU.apply(p.reshape(T * B, I), w_ih, w_hh, b_ih, b_hh, copies, T, B, H, ndir, x_length)
    return logical3(y.reshape(T, B, H))


# ---------------------------------------------------------------------------------------------- array manipulation (views)
def reshape(x, shape):
    """chainer.functions.reshape.  Merging (C, H) of a physical image -- run/ctc/sru/model.py:114
    ``reshape(out, (B, -1, T))`` -- is a free view; the merged feature order is (h, c) instead of (c, h), a fixed
    permutation of the next layer's input columns."""
    shape = tuple(shape)
    if x.dim() == 4 and len(shape) == 3:
        p = x.permute(3, 0, 2, 1)
        B, C, H, T = x.shape
        if p.is_contiguous() and shape[0] == B and shape[2] == T and shape[1] in (-1, C * H):
            return p.reshape(T, B, H * C).permute(1, 2, 0)
    return x.reshape(shape)


def argmax(x, axis=2):
    """xp.argmax(y_batch.data, axis=2) of the evaluation loop (run/ctc/cnn/dev.py:106): x is the (B, T, V) float32 view
    a model returns with split_into_variables=False; -> (B, T) int32 ids, first maximum on ties."""
    assert x.dim() == 3 and axis in (2, -1)
    tbv = x.detach().permute(1, 0, 2)
    if not tbv.is_contiguous():
        tbv = tbv.contiguous()
    return _ops.argmax_rows(tbv.float() if tbv.dtype != torch.float32 else tbv)


def swapaxes(x, a, b):
    return x.transpose(a, b)


def transpose(x, axes):
    return x.permute(*axes)


def squeeze(x, axis=None):
    return x.squeeze() if axis is None else x.squeeze(axis)


def expand_dims(x, axis):
    return x.unsqueeze(axis)


def split_axis(x, indices_or_sections, axis):
    if isinstance(indices_or_sections, int):
        return torch.chunk(x, indices_or_sections, dim=axis)
    return torch.tensor_split(x, list(indices_or_sections), dim=axis)


def flatten(x):
    return x.reshape(-1)


def broadcast_to(x, shape):
    return x.expand(*shape)


def tile(x, reps):
    return x.repeat(*reps) if not isinstance(reps, int) else x.repeat(reps)


def rollaxis(x, axis, start=0):
    return torch.movedim(x, axis, start if start <= axis else start - 1)


from .loss.ctc import connectionist_temporal_classification, gram_ctc  # noqa: E402,F401
