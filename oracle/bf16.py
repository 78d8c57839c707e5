"""Rounding-matched CPU oracle: the fp32 restatement with bf16 roundings at the points where the HIP path rounds.
TEST INFRASTRUCTURE -- see oracle/__init__.py.

The fp32 oracles (oracle/model.py, oracle/cnn.py) differ from the device by ~40 bf16 roundings per step, so a gate against them
has to be loose enough to hide a small bug ("What's weak" 1).  This module restates the SAME arithmetic with a
round-to-nearest-even to bf16 wherever the device stores bf16 (and nowhere else), accumulating in float32 as the MFMA does:

  forward   every activation tensor an operator writes (convolution / dense outputs after the bias add, activation / GLU / residual
            add outputs, bf16 layer-norm outputs, the GRU's input projections when the kernel streams them in bf16, the GRU layer
            output hf + hb); MFMA operands (weights cast once, the bf16 copy of h_{t-1} the recurrence multiplies); NOT the f32
            hidden state, saved gates, layer-norm statistics, logits, softmax, CTC lattice
  backward  every activation gradient an operator writes (backward-data GEMMs, activation / GLU / layer-norm dx, the fused
            LayerNorm + CTC dx handed to the logit projection, the per-step gate gradients dgi / dgh of the recurrence and -- in the
            partial-sum kernel -- the per-producer partial sums of dgh . W_hh); sums of two gradient tensors at a fork (torch adds
            bf16 tensors: one more rounding); NOT weight / bias gradients (float32 accumulation of bf16 products)

What stays different is float32 summation order, the fast exp / rcp of the gate math, and the CTC lattice (float64 on the device,
torch's float32 ctc_loss here): ~1e-5, two orders below one bf16 ulp (3.9e-3 relative).
Reference arithmetic: asr/nn/nn.py (layers), asr/nn/layernorm.py:33-61, run/ctc/cnn/model.py:11-332, chainer's GRU / conv /
CTC as restated in oracle/nn.py and oracle/model.py (unpinned boundary, see oracle/__init__.py).
"""
import os

import torch
import torch.nn.functional as F


# the 16-bit format the device rounds to: bfloat16 (the default library) or IEEE half when the half build is under test (ASR_ACT=f16,
# csrc/common.hpp) -- the same environment variable that selects the library
ACT = torch.float16 if os.environ.get("ASR_ACT", "").lower() in ("f16", "fp16", "float16", "half") else torch.bfloat16


def rnd(x):
    """float32 -> nearest-even 16-bit activation format -> float32 (what v_cvt_pk_bf16_f32 / (__bf16)f, or v_cvt_f16_f32 in the half
    build, does on the device)"""
    return x.to(ACT).to(torch.float32)


class _RoundFB(torch.autograd.Function):
    """forward: the tensor is stored in bf16; backward: the (summed) gradient arriving here is a bf16 tensor"""

    @staticmethod
    def forward(ctx, x):
        return rnd(x)

    @staticmethod
    def backward(ctx, g):
        return rnd(g)


class _RoundB(torch.autograd.Function):
    """identity forward; backward: the gradient this consumer writes for its input is stored in bf16"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return rnd(g)


def out(x, on=True):
    return _RoundFB.apply(x) if on else x


def inp(x, on=True):
    return _RoundB.apply(x) if on else x


def weight(w, on=True):
    """the bf16 compute copy of a float32 master weight (gradient flows to the master unrounded: float32 accumulation)"""
    if not on:
        return w
    return w + (rnd(w.detach()) - w.detach())


# ------------------------------------------------------------------------------------------------------------------ GRU
class GRUMatched(torch.autograd.Function):
    """(Bi)GRU layer with the device's rounding points (csrc/gru.hip): cuDNN gate convention, directions summed.

    x (T, B, I) bf16-valued; w_ih (ndir, 3H, I), w_hh (ndir, 3H, H), b_ih, b_hh (ndir, 3H) float32 masters.
    gi_bf16: the input projections are written in bf16 (asr_gru_fwd_accepts_bf16_gi); gates_f16: the gates saved for the backward
    pass are kept in IEEE half (asr_gru_gates_f16_ok).  ps_units: 32 when the partial-sum backward
    kernel serves the shape (H % 128 == 0: every 32-unit producer publishes its share of dgh . W_hh rounded to bf16), else 0.
    x_len (B) or None: rows are live for t < x_len[b]; beyond, the state is frozen (the reverse direction therefore starts from a
    zero state at x_len[b] - 1), the output is zero and no gradient flows -- NStepBiGRU's per-sequence lengths (asr/nn/nn.py:3).
    """

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, gi_bf16, ps_units, x_len, gates_f16=False):
        T, B, I = x.shape
        ndir, H = w_hh.shape[0], w_hh.shape[2]
        wih, whh = rnd(w_ih), rnd(w_hh)
        live = None
        if x_len is not None:
            live = (torch.arange(T).reshape(T, 1) < x_len.reshape(1, B).long()).reshape(T, B, 1)
        hsum = torch.zeros(T, B, H)
        saved = []
        for d in range(ndir):
            gi = x.reshape(T * B, I) @ wih[d].t() + b_ih[d]
            if gi_bf16:
                gi = rnd(gi)
            gi = gi.reshape(T, B, 3 * H)
            h = torch.zeros(B, H)
            R, Z, N, Q, HP = (torch.empty(T, B, H) for _ in range(5))
            for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
                gh = rnd(h) @ whh[d].t() + b_hh[d]
                r = torch.sigmoid(gi[t, :, :H] + gh[:, :H])
                z = torch.sigmoid(gi[t, :, H:2 * H] + gh[:, H:2 * H])
                n = torch.tanh(gi[t, :, 2 * H:] + r * gh[:, 2 * H:])
                hn = (1.0 - z) * n + z * h
                if live is not None:
                    hn = torch.where(live[t], hn, h)
                R[t], Z[t], N[t], Q[t], HP[t] = r, z, n, gh[:, 2 * H:], h
                h = hn
                hsum[t] += hn
            if gates_f16:       # the default kernel pair keeps the saved gates in IEEE half (asr_gru_gates_f16_ok); the state stays float32
                R, Z, N, Q = (a.to(torch.float16).to(torch.float32) for a in (R, Z, N, Q))
            saved.append((R, Z, N, Q, HP))
        y = rnd(hsum)
        if live is not None:
            y = y * live
        ctx.save_for_backward(x, wih, whh)
        ctx.saved_gates = saved
        ctx.meta = (T, B, I, H, ndir, int(ps_units or 0), live)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wih, whh = ctx.saved_tensors
        T, B, I, H, ndir, ps, live = ctx.meta
        dy = rnd(dy)
        if live is not None:
            dy = dy * live
        gx_acc = torch.zeros(T * B, I)
        gw_ih, gw_hh = torch.zeros(ndir, 3 * H, I), torch.zeros(ndir, 3 * H, H)
        gb_ih, gb_hh = torch.zeros(ndir, 3 * H), torch.zeros(ndir, 3 * H)
        x2 = x.reshape(T * B, I)
        if ps:      # producer p owns units [p ps, (p + 1) ps) of all three gates: W_hh rows as (gate, producer, unit, column)
            P = H // ps
            whh_p = [whh[d].reshape(3, P, ps, H) for d in range(ndir)]
        for d in range(ndir):
            R, Z, N, Q, HP = ctx.saved_gates[d]
            dgi, dgh = torch.zeros(T, B, 3 * H), torch.zeros(T, B, 3 * H)
            carry, rec = torch.zeros(B, H), torch.zeros(B, H)
            for t in (range(T - 1, -1, -1) if d == 0 else range(T)):
                dh = dy[t] + carry + rec
                r, z, n, q, hp = R[t], Z[t], N[t], Q[t], HP[t]
                dan = dh * ((1.0 - z) * (1.0 - n * n))
                daz = dh * ((hp - n) * z * (1.0 - z))
                dq = dan * r
                dar = dan * (q * r * (1.0 - r))
                gi_t = rnd(torch.cat([dar, daz, dan], dim=1))
                gh_t = rnd(torch.cat([dar, daz, dq], dim=1))
                carry_n = dh * z
                if live is not None:
                    gi_t, gh_t = gi_t * live[t], gh_t * live[t]
                    carry_n = torch.where(live[t], carry_n, dh)
                dgi[t], dgh[t] = gi_t, gh_t
                carry = carry_n
                if ps:      # every producer's share of dgh . W_hh is published in bf16, the consumer adds them up in float32
                    rec = rnd(torch.einsum("bgpu,gpuh->pbh", gh_t.reshape(B, 3, P, ps), whh_p[d])).sum(dim=0)
                else:
                    rec = gh_t @ whh[d]
            gx_acc += dgi.reshape(T * B, 3 * H) @ wih[d]
            gw_ih[d] = dgi.reshape(T * B, 3 * H).t() @ x2
            gw_hh[d] = dgh.reshape(T * B, 3 * H).t() @ rnd(HP).reshape(T * B, H)
            gb_ih[d] = dgi.sum(dim=(0, 1))
            gb_hh[d] = dgh.sum(dim=(0, 1))
        gx = rnd(gx_acc).reshape(T, B, I)
        return gx, gw_ih, gw_hh, gb_ih, gb_hh, None, None, None, None


def gru_f32(x, w_ih, w_hh, b_ih, b_hh, x_len=None):
    """the float32 statement of the same layer: torch's CPU GRU per direction, directions summed; with x_len through
    pack_padded_sequence (outputs zero beyond the length, the reverse direction starting at x_len[b] - 1)"""
    T, B, I = x.shape
    ndir, H = w_hh.shape[0], w_hh.shape[2]
    if x_len is None:
        outs = []
        for d in range(ndir):
            xs = x if d == 0 else x.flip(0)
            hs, _ = torch._VF.gru(xs, torch.zeros(1, B, H), [w_ih[d], w_hh[d], b_ih[d], b_hh[d]], True, 1, 0.0, False, False, False)
            outs.append(hs if d == 0 else hs.flip(0))
        return outs[0] if ndir == 1 else outs[0] + outs[1]
    flat = []
    for d in range(ndir):
        flat += [w_ih[d], w_hh[d], b_ih[d], b_hh[d]]
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, x_len.long().cpu(), enforce_sorted=False)
    data, batch_sizes, sorted_idx, unsorted_idx = packed
    h0 = torch.zeros(ndir, B, H)
    res, _ = torch._VF.gru(data, batch_sizes, h0, flat, True, 1, 0.0, False, ndir == 2)
    out_packed = torch.nn.utils.rnn.PackedSequence(res, batch_sizes, sorted_idx, unsorted_idx)
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(out_packed, total_length=T)
    return y if ndir == 1 else y[..., :H] + y[..., H:]


def gru_f32_state(x, w_ih, w_hh, b_ih, b_hh, h0, x_len=None):
    """gru_f32 with an initial state h0 (ndir, B, H): -> (y (T, B, H) directions summed, hy (ndir, B, H)) -- NStepGRU's hx / hy"""
    T, B, I = x.shape
    ndir, H = w_hh.shape[0], w_hh.shape[2]
    flat = []
    for d in range(ndir):
        flat += [w_ih[d], w_hh[d], b_ih[d], b_hh[d]]
    if x_len is None:
        res, hy = torch._VF.gru(x, h0, flat, True, 1, 0.0, False, ndir == 2, False)
        return (res if ndir == 1 else res[..., :H] + res[..., H:]), hy
    packed = torch.nn.utils.rnn.pack_padded_sequence(x, x_len.long().cpu(), enforce_sorted=False)
    data, batch_sizes, sorted_idx, unsorted_idx = packed
    res, hy = torch._VF.gru(data, batch_sizes, h0.index_select(1, sorted_idx), flat, True, 1, 0.0, False, ndir == 2)
    out_packed = torch.nn.utils.rnn.PackedSequence(res, batch_sizes, sorted_idx, unsorted_idx)
    y, _ = torch.nn.utils.rnn.pad_packed_sequence(out_packed, total_length=T)
    return (y if ndir == 1 else y[..., :H] + y[..., H:]), hy.index_select(1, unsorted_idx)


def gru(x, w_ih, w_hh, b_ih, b_hh, x_len=None, matched=False, gi_bf16=True, ps_units=None, gates_f16=False):
    if not matched:
        return gru_f32(x, w_ih, w_hh, b_ih, b_hh, x_len)
    if ps_units is None:
        ps_units = 32 if w_hh.shape[2] % 128 == 0 else 0
    return GRUMatched.apply(x, w_ih, w_hh, b_ih, b_hh, bool(gi_bf16), int(ps_units), x_len, bool(gates_f16))


# ------------------------------------------------------------------------------------------------------------------ helpers
def linear(h, W, b, on, f32_out=False, bias_grad_unrounded=False):
    """dense / 1x1 convolution on rows: bf16 operands, float32 accumulation + bias, bf16 store -- or, f32_out (the logits), a
    float32 store whose incoming gradient is handed over in bf16.  bias_grad_unrounded: the bias gradient is the column sum of the
    UNROUNDED output gradient (the fused LayerNorm + CTC sweep forms it in float32 before dx is rounded for the hand-over:
    csrc/ctc_ln.hip, template parameter XS); otherwise the column sum of the bf16 gradient (asr_colsum_acc)"""
    y = F.linear(inp(h, on), weight(W, on))
    if f32_out:
        return (inp(y, on) + b) if bias_grad_unrounded else inp(y + b, on)
    return out(y + b, on)


def layer_norm_rows(h, gamma, beta):
    """per-row normalisation over the last axis, no epsilon (asr/nn/layernorm.py:33-48), scale / bias per column.  A row of
    zero variance (all-zero frames beyond an utterance's length with zero biases) is NaN in the forward pass, as in the
    reference, and receives / contributes no gradient (the HIP backward kernels skip such rows: csrc/layernorm.hip)"""
    mean = h.mean(dim=-1, keepdim=True)
    diff = h - mean
    var = (diff * diff).mean(dim=-1, keepdim=True)
    ok = var > 0
    y = diff / torch.sqrt(torch.where(ok, var, torch.ones_like(var))) * gamma + beta
    return torch.where(ok, y, torch.full_like(y, float("nan")))
