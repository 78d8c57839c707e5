"""BASELINE configs[1]'s model (2 x conv + 4 x BiGRU-512 + dense + LayerNorm + CTC) against the rounding-matched oracle

(a) LAYER BY LAYER, teacher-forced -- the recurrent twin of test_model_gpu.py::test_cnn_recipes_layer_by_layer_against_the_matched_oracle
    (which follows the reference's own method, asr/nn/test_layernorm.py:45-74: a tight forward + gradient check per operator): every unit
    of the DS2 stack -- each convolution, each fused Maxout + MaxPooling pair, EACH BiGRU LAYER at H = 512, each dense projection and
    maxout, the logit projection + LayerNormalization + CTC head -- is given the DEVICE's own input and the DEVICE's own output gradient;
    its output, its input gradient and its parameter gradients are compared with oracle/bf16.py's statement of that ONE unit.  Nothing
    accumulates across layers, so the bars are tight and say "every layer is right"; what is left inside a recurrent layer is the
    amplification of float32 summation-order differences over its own T dependent steps.
(b) END TO END at the bench's full length: one forward + CTC + backward of the whole model at T = 1000 (B = 4 keeps the CPU oracle under a
    minute), gated like bench.py gates its own line.

Why both: two correct bf16 implementations drift apart with depth (one flipped rounding is amplified by every layer behind it), so the
end-to-end bar cannot be tight AND depth independent.  (a) is the tight one; (b) pins the composition.  VERDICT r3 next 4."""
import numpy as np
import pytest
import torch

from asr._lib import act_dtype as _act_dtype      # bfloat16, or float16 when the half build is under test (ASR_ACT=f16)

from oracle import bf16 as Q
from oracle import model as omodel
from oracle import nn as onn

pytestmark = pytest.mark.gpu
BF16 = _act_dtype()

# per-unit bars (relative L2), ~2 x what is measured on the device (printed by the test, run with -s).  Measured, B = 20 / 17, T = 150 / 120:
#   convolutions, dense projections: forward 2.6e-5, input gradient 3.2e-5, parameter gradients 1e-6 (float32 summation order)
#   fused Maxout + MaxPooling, Maxout: exact
#   a BiGRU-512 layer: output 4.9e-4, parameter gradients 1.7e-4, input gradient 2.0e-3 -- inside ONE layer the per-step gate gradients are
#     bf16 tensors that went through up to T dependent steps; dx = dgi . W_ih sums 3072 of them per element and is rounded to bf16 again,
#     dW = dgi^T x averages the same noise over T x B rows
#   logit projection + LayerNormalization + CTC: logits 1.4e-7, loss 7e-8, parameter gradients 1.7e-4, input gradient 1.0e-3 (bf16 dx
#     of a 3000-term sum per element)
BARS = {("conv", "forward"): 1e-4, ("conv", "input gradient"): 1e-4, ("conv", "parameter gradient"): 1e-5,
        ("dense", "forward"): 1e-4, ("dense", "input gradient"): 1e-4, ("dense", "parameter gradient"): 1e-5,
        ("pool", "forward"): 1e-7, ("pool", "input gradient"): 1e-7, ("maxout", "forward"): 1e-7, ("maxout", "input gradient"): 1e-7,
        ("gru", "forward"): 1e-3, ("gru", "input gradient"): 4e-3, ("gru", "parameter gradient"): 4e-4,
        ("head", "logits"): 1e-6, ("head", "loss"): 1e-6, ("head", "input gradient"): 2e-3, ("head", "parameter gradient"): 4e-4}


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _build(device, V, seed):
    from asr.model import ds2
    torch.manual_seed(seed)
    cfg = ds2.configure()                   # configs[1]: ndim_conv 64, 4 x BiGRU-512, dense 320
    cfg.vocab_size = V
    model = ds2.Model(cfg).to_gpu()
    return cfg, model


def _randomise_biases(model, seed):
    """zero biases + all-zero frames beyond an utterance's length make the epsilon-free LayerNormalization 0/0 there (reference
    behaviour, covered by test_gru_lengths_gpu.py); this test wants finite rows everywhere"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith(".b") or name.endswith(".beta"):
                p.copy_((torch.rand(p.shape, generator=g) * 0.2 - 0.1).to(p.device))


def _trace_ds2(model, xd, x_length):
    """Model.__call__ of asr/model/ds2.py unit by unit, recording every unit's input, output and (by hook) output gradient"""
    from asr import functions as F
    from asr import nn
    from asr.nn import nn as nnmod
    from asr.model._acoustic import split_output
    names = {id(m): n for n, m in model.named_modules()}
    rec = []

    def unit(kind, layer, fn, x):
        y = fn(x)
        e = dict(kind=kind, name=names.get(id(layer)) if layer is not None else None, xin=x.detach(), xout=y.detach(), gout=None)
        if y.requires_grad:
            y.register_hook(lambda g, e=e: e.__setitem__("gout", g.detach().clone()))
        rec.append(e)
        return y

    def run_list(layers, x, lengths=None):
        i = 0
        while i < len(layers):
            layer = layers[i]
            j = nnmod._fusable_pool(layers, i) if x.dim() == 4 else -1
            if j > 0:
                ks = layers[j].ksize
                k = ks[0] if isinstance(ks, (tuple, list)) else ks
                x = unit("pool%d" % k, None, lambda t, k=k, sole=(i > 0): F.maxout_max_pooling(t, k, sole_consumer=sole), x)
                i = j + 1
                continue
            tname = type(layer).__name__
            if tname == "Dropout":
                x = layer(x)
            elif isinstance(layer, (nn.GRU, nn.BiGRU)):
                x = unit("gru", layer, (lambda t, l=layer: l(t, lengths)) if lengths is not None else layer, x)
            elif tname == "Maxout":
                x = unit("maxout", None, layer, x)
            else:
                assert tname.endswith("Convolution2D") or tname.endswith("Convolution1D"), tname
                x = unit("conv" if tname.endswith("Convolution2D") else "dense", layer, layer, x)
            i += 1
        return x
    B, T = xd.shape[0], xd.shape[3]
    h = run_list(model.conv_blocks.layers, xd)
    h = F.reshape(h, (B, -1, T))
    h = run_list(model.rnn_blocks.layers, h, x_length)
    dl = model.dense_blocks.layers
    h = run_list(dl[:-2], h)
    head_in = h
    out = dl[-1](dl[-2](h))
    ys = split_output(out, B, T, True)
    rec.append(dict(kind="head", name=(names[id(dl[-2])], names[id(dl[-1])]), xin=head_in.detach(), xout=None, gout=None))
    return ys, rec


@pytest.mark.parametrize("B,T,ragged", [(20, 150, True), (17, 120, False)])       # (B > 16: the default kernel pair, as at B = 32)
def test_ds2_stack_layer_by_layer_against_the_matched_oracle(device, B, T, ragged):
    from asr import _ops
    from asr import functions as F
    from asr.loss import connectionist_temporal_classification
    V, H, ndir = 3000, 512, 2
    cfg, model = _build(device, V, seed=11)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=5, Lmax=20, seed=3, ragged=ragged)
    xd = x.to(device)
    with torch.no_grad():
        model(xd)                       # lazily sized parameters
    _randomise_biases(model, 5)
    before = _ops.CALLS.get("layernorm_ctc_bwd", 0)
    ys, rec = _trace_ds2(model, xd, x_len.to(device) if ragged else None)
    loss = connectionist_temporal_classification(ys, labels.to(device), 0, x_len.to(device), l_len.to(device))
    loss.backward()
    F.join_side_stream()
    torch.cuda.synchronize()
    _ops.gru_check_sync()
    fused = _ops.CALLS.get("layernorm_ctc_bwd", 0) > before
    assert fused                        # V % 4 == 0: the CTC gradient is formed inside LayerNormalization's backward
    P = {n: p.detach().float().cpu() for n, p in model.named_parameters()}
    G = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}
    gi_bf16 = _ops.gru_gi_dtype(T, B, H, ndir) == BF16
    gates_f16 = _ops.gru_gates_f16(T, B, H, ndir)
    assert gi_bf16 and gates_f16        # the default kernel pair serves this shape
    worst = {}
    checked = set()

    failures = []

    def note(kind, what, err, tag):
        key = (kind, what)
        bar = BARS[("pool" if kind.startswith("pool") else kind, what)]
        if err > worst.get(key, (-1.0, ""))[0]:
            worst[key] = (err, tag)
        if not err < bar:
            failures.append((kind, what, tag, err, bar))

    def param(name):
        return P[name].clone().requires_grad_(True)

    for k, e in enumerate(rec):
        kind = e["kind"]
        xin = e["xin"].float().cpu()
        if k == 0:
            xin = Q.rnd(xin)                                    # the first layer packs the loader's float32 minibatch to bf16
        xin = xin.clone().requires_grad_(k > 0)
        ps = {}
        if kind == "conv":
            ps = {n: param(e["name"] + "." + n) for n in ("W", "b")}
            y = Q.out(onn.conv2d_causal(Q.inp(xin), Q.weight(ps["W"]), ps["b"], 0))
        elif kind.startswith("pool"):
            y = Q.out(onn.maxpool_h(onn.maxout2(xin), int(kind[4:])))
        elif kind == "gru":
            ps = {n: param(e["name"] + "." + n) for n in ("w_ih", "w_hh", "b_ih", "b_hh")}
            y = Q.gru(xin.permute(2, 0, 1), ps["w_ih"], ps["w_hh"], ps["b_ih"], ps["b_hh"], x_len if ragged else None, True, gi_bf16, None,
                      gates_f16).permute(1, 2, 0)
        elif kind == "dense":
            ps = {n: param(e["name"] + "." + n) for n in ("W", "b")}
            y = Q.linear(xin.permute(2, 0, 1), ps["W"][:, :, 0], ps["b"], True).permute(1, 2, 0)
        elif kind == "maxout":
            Bx, D, Tx = xin.shape
            y = Q.out(xin.reshape(Bx, D // 2, 2, Tx).max(dim=2)[0])
        else:
            assert kind == "head"
            proj, norm = e["name"]
            ps = {"W": param(proj + ".W"), "b": param(proj + ".b"), "gamma": param(norm + ".norm.gamma"), "beta": param(norm + ".norm.beta")}
            lg = Q.layer_norm_rows(Q.linear(xin.permute(2, 0, 1), ps["W"][:, :, 0], ps["b"], True, f32_out=True, bias_grad_unrounded=True),
                                   ps["gamma"], ps["beta"])
            logits = torch.stack(tuple(ys)).detach().float().cpu()
            live = (torch.arange(T).reshape(T, 1) < x_len.reshape(1, B).long()).reshape(T, B, 1).expand_as(logits)
            note("head", "logits", _rel(logits[live], lg.detach()[live]), "logit projection + LayerNormalization")
            loss_ref = omodel.ctc_mean_loss(lg, labels, x_len, l_len)
            note("head", "loss", abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), "CTC")
            loss_ref.backward()
            y = None
        tag = "%s %s" % (kind, e["name"])
        if y is not None:
            note(kind, "forward", _rel(e["xout"].float().cpu(), y.detach()), tag)
            assert e["gout"] is not None, tag
            y.backward(e["gout"].float().cpu())
        if k > 0:
            gprev = rec[k - 1]["gout"]
            assert gprev is not None, tag
            gprev = gprev.float().cpu()
            if gprev.dim() == 4 and xin.dim() == 3:         # the conv stack's (B, C, H, T) merged to (B, H C, T): feature order (h, c), see ds2.py
                gprev = gprev.permute(0, 2, 1, 3).reshape(xin.shape)
            note(kind, "input gradient", _rel(gprev, Q.rnd(xin.grad)), tag)
        for n, v in ps.items():
            full = (e["name"][1] + ".norm." + n) if (kind == "head" and n in ("gamma", "beta")) else ((e["name"][0] if kind == "head" else e["name"]) + "." + n)
            note(kind, "parameter gradient", _rel(G[full], v.grad), full)
            checked.add(full)
    print("DS2 layer by layer, B=%d T=%d %s:" % (B, T, "ragged" if ragged else "full length"))
    for (kind, what), (err, tag) in sorted(worst.items()):
        print("   %-8s %-18s worst %.2e  (%s)" % (kind, what, err, tag))
    assert not failures, failures
    assert checked == set(G), sorted(set(G) - checked)


def test_whole_configs1_model_step_at_full_length(device):
    """(b): BASELINE configs[1] end to end at T = 1000 -- 2 x conv + 4 x BiGRU-512 + dense + LayerNorm + CTC, one forward + backward, B = 4
    utterances (labels 40..120, V = 3000), against the rounding-matched oracle: loss 1e-4, logits 1e-2, every parameter gradient at
    FULL_GRAD (2 x the 1.1e-2 this composition measures; see the module docstring for why it is not the per-layer bar), and loosely
    against the plain float32 oracle (the price of bf16 activations, reported)."""
    from asr import _ops
    from asr import functions as F
    from asr.loss import connectionist_temporal_classification
    FULL_GRAD = 2.5e-2
    B, T, V, H = 4, 1000, 3000, 512
    cfg, model = _build(device, V, seed=0)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, seed=0)
    xd = x.to(device)
    with torch.no_grad():
        model(xd)
    ys = model(xd)
    loss = connectionist_temporal_classification(ys, labels.to(device), 0, x_len.to(device), l_len.to(device))
    loss.backward()
    F.join_side_stream()
    torch.cuda.synchronize()
    _ops.gru_check_sync()
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    logits = torch.stack(tuple(ys)).detach().float().cpu()
    for matched in (True, False):
        ref = omodel.DS2Oracle(state, cfg.num_conv_layers, cfg.num_rnn_layers, True, matched=matched,
                               gi_bf16=_ops.gru_gi_dtype(T, B, H, 2) == BF16, fused_logit_bias=True, gates_f16=_ops.gru_gates_f16(T, B, H, 2))
        lr = ref(x)
        loss_ref = omodel.ctc_mean_loss(lr, labels, x_len, l_len)
        loss_ref.backward()
        errs = {name: _rel(p.grad.cpu(), ref.g(name).grad) for name, p in model.named_parameters()}
        worst = max(errs, key=errs.get)
        lrel = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
        print("configs[1] T=1000 B=4 matched=%s: loss %.2e logits %.2e worst gradient %.2e (%s)" % (matched, lrel, _rel(logits, lr.detach()), errs[worst], worst))
        if matched:
            assert lrel < 1e-4
            assert _rel(logits, lr.detach()) < 1e-2
            assert errs[worst] < FULL_GRAD, (worst, errs[worst])
        else:
            assert lrel < 2e-2 and errs[worst] < 0.25, (lrel, worst, errs[worst])
