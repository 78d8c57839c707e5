"""CPU oracle of the Deep-Speech-2-style CTC model and its train step (torch-CPU fp32).  TEST INFRASTRUCTURE.

Same architecture and parameter tensors as chainer-speech-recognition_amd/asr/model/ds2.py, expressed with stock
torch-CPU operators in the reference's logical layouts.  Used (a) by tests as the checker of the HIP train step and
(b) by bench.py's ``cpu_baseline`` leg as the timed CPU restatement of the path (the literal Chainer path cannot run:
Chainer is not installed and the GPU box has no /root/reference).
conv / max-pool / maxout / GRU / Adam are Chainer's in the reference -> parity unpinned at that boundary (oracle/__init__.py);
layer-norm follows asr/nn/layernorm.py (no epsilon); the CTC term is torch's ctc_loss, itself checked against
oracle.ctc (tests/test_oracle_ctc.py), with Chainer's normalisation (mean over the batch of -log p).
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import nn as onn


class DS2Oracle(torch.nn.Module):
    def __init__(self, state, num_conv_layers, num_rnn_layers, bidirectional=True, matched=False, gi_bf16=True, ps_units=None,
                 fused_logit_bias=False, gates_f16=False):
        """state: dict name -> float32 CPU tensor with the names of asr.model.ds2.Model.state_dict().
        matched: restate the step with bf16 roundings where the HIP path rounds (oracle/bf16.py) -- gi_bf16 / ps_units say which
        recurrence kernels serve the shape (asr_gru_fwd_accepts_bf16_gi; partial-sum backward at H % 128 == 0), fused_logit_bias
        that the logit projection's bias gradient comes out of the fused LayerNorm + CTC sweep (V % 4 == 0) in float32."""
        super().__init__()
        self.nconv, self.nrnn, self.ndir = num_conv_layers, num_rnn_layers, 2 if bidirectional else 1
        self.matched, self.gi_bf16, self.ps_units, self.fused_logit_bias = bool(matched), bool(gi_bf16), ps_units, bool(fused_logit_bias)
        self.gates_f16 = bool(gates_f16)
        self.p = torch.nn.ParameterDict({k.replace(".", "__"): torch.nn.Parameter(v.clone().float()) for k, v in state.items()})

    def g(self, name):
        return self.p[name.replace(".", "__")]

    def forward(self, x, x_len=None):
        """x (B, 3, 40, T) -> logits (T, B, V).  x_len: per-utterance frame counts for the recurrent layers (NStepBiGRU semantics:
        the reverse direction of utterance b starts at x_len[b] - 1, outputs beyond it are zero); None = the padded block."""
        from . import bf16 as Q
        on = self.matched
        pools = [3] + [2] * (self.nconv - 1)
        h = Q.rnd(x) if on else x
        for i, pool in enumerate(pools):
            W, b = self.g("conv_blocks._sequential_%d.W" % (4 * i)), self.g("conv_blocks._sequential_%d.b" % (4 * i))
            h = Q.out(onn.conv2d_causal(Q.inp(h, on), Q.weight(W, on), b, 0), on)
            h = Q.out(onn.maxpool_h(onn.maxout2(h), pool), on)
        B, C, H, T = h.shape
        h = h.permute(3, 0, 2, 1).reshape(T, B, H * C)          # feature order (h, c): see ds2.py / functions.reshape
        for i in range(self.nrnn):
            pre = "rnn_blocks._sequential_%d." % (2 * i)
            w_ih, w_hh, b_ih, b_hh = (self.g(pre + n) for n in ("w_ih", "w_hh", "b_ih", "b_hh"))
            h = Q.gru(h, w_ih, w_hh, b_ih, b_hh, x_len, on, self.gi_bf16, self.ps_units, self.gates_f16)
        for j, idx in enumerate((0, 3)):
            W, b = self.g("dense_blocks._sequential_%d.W" % idx), self.g("dense_blocks._sequential_%d.b" % idx)
            h = Q.linear(h, W[:, :, 0], b, on)
            h = Q.out(h.reshape(T, B, -1, 2).max(dim=3)[0], on)
        W, b = self.g("dense_blocks._sequential_6.W"), self.g("dense_blocks._sequential_6.b")
        h = Q.linear(h, W[:, :, 0], b, on, f32_out=True, bias_grad_unrounded=self.fused_logit_bias)
        gamma, beta = self.g("dense_blocks._sequential_7.norm.gamma"), self.g("dense_blocks._sequential_7.norm.beta")
        return Q.layer_norm_rows(h, gamma, beta)


def ctc_mean_loss(logits, labels, x_len, l_len, blank=0):
    """Chainer's normalisation: mean over utterances of -log p(label | x)."""
    T = logits.shape[0]
    live = (torch.arange(T).reshape(T, 1) < x_len.reshape(1, -1).long()).unsqueeze(2)
    logits = torch.where(live, logits, torch.zeros_like(logits))      # frames beyond the length never reach the loss (may be NaN)
    lp = torch.log_softmax(logits, dim=2)
    loss = F.ctc_loss(lp, labels.long(), x_len.long(), l_len.long(), blank=blank, reduction="none", zero_infinity=False)
    return loss.mean()


def train_step(model, state_m, state_v, step, x, labels, x_len, l_len, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8,
               decay=1e-5, clip=1.0):
    """forward + CTC + backward + GradientClipping(1) + WeightDecay(1e-5) + Adam (run/ctc/cnn/train.py:142-147,190-200)."""
    for p in model.parameters():
        p.grad = None
    loss = ctc_mean_loss(model(x), labels, x_len, l_len)
    loss.backward()
    params = list(model.parameters())
    norm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params)).item()
    rate = clip / norm if clip > 0 else 1.0
    lr = alpha * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    with torch.no_grad():
        for i, p in enumerate(params):
            g = p.grad * (rate if rate < 1 else 1.0) + decay * p
            state_m[i] += (1 - beta1) * (g - state_m[i])
            state_v[i] += (1 - beta2) * (g * g - state_v[i])
            p -= lr * state_m[i] / (torch.sqrt(state_v[i]) + eps)
    return loss.item(), norm


def synthetic_batch(B, T, V, Lmin=40, Lmax=120, seed=0, ragged=False, nmel=40):
    """x ~ N(0,1) (B,3,nmel,T) f32, labels U{1..V-1}, L ~ U{Lmin..Lmax}, padded with 0."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, 3, nmel, T, generator=g)
    l_len = torch.randint(Lmin, Lmax + 1, (B,), generator=g, dtype=torch.int32)
    labels = torch.randint(1, V, (B, Lmax), generator=g, dtype=torch.int32)
    for b in range(B):
        labels[b, l_len[b]:] = 0
    if ragged:
        x_len = torch.randint(int(0.6 * T), T + 1, (B,), generator=g, dtype=torch.int32)
    else:
        x_len = torch.full((B,), T, dtype=torch.int32)
    return x, labels, x_len, l_len
