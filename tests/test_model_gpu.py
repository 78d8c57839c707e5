"""The DS2-style model and its train step on the HIP path against the torch-CPU oracle (same parameters, same batch).

Activations and MFMA operands are bf16 (BASELINE.json config: "1xMI355X bf16"), so the model-level tolerances are
bf16 tolerances; the CTC loss/gradient itself is checked to 1e-4 in tests/test_ctc_gpu.py on identical logits."""
import numpy as np
import pytest
import torch

from asr._lib import act_dtype as _act_dtype      # bfloat16, or float16 when the half build is under test (ASR_ACT=f16)

from oracle import model as omodel

pytestmark = pytest.mark.gpu
BF16 = _act_dtype()
# the IEEE-half build (ASR_ACT=f16, tests/test_f16_gpu.py) is only usable with a scaled backward seed (Optimizer.loss_scaling): the recipe
# tests seed both sides' backward passes with it
LOSS_SCALE = 1024.0 if BF16 is torch.float16 else 1.0


def _seeded_backward(loss, scale=None):
    scale = LOSS_SCALE if scale is None else scale
    if scale == 1.0:
        loss.backward()
    else:
        loss.backward(gradient=torch.full_like(loss, scale))


def _cos(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return (a @ b / (a.norm() * b.norm() + 1e-30)).item()


def _build(device, V=29, conv=16, rnn=64, dense=32, nrnn=2, bidir=True, seed=0):
    from asr.model import ds2
    torch.manual_seed(seed)
    cfg = ds2.configure()
    cfg.vocab_size, cfg.ndim_conv, cfg.ndim_rnn, cfg.ndim_dense, cfg.num_rnn_layers, cfg.bidirectional = V, conv, rnn, dense, nrnn, bidir
    model = ds2.Model(cfg)
    model.to_gpu()
    return cfg, model


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# Gates.  MATCHED: the oracle rounds to bf16 where the device does (oracle/bf16.py), so what is left is float32 summation order, the
# fast exp / rcp of the gate math and float32-vs-float64 CTC -- a dropped tap, a wrong pad or a missing term shows at once.
# FP32: the plain float32 oracle; the distance to it is the price of bf16 activations (reported, loosely bounded).
MATCHED_GRAD_REL_L2 = 5e-3
MATCHED_GRAD_RNN512 = 2.5e-2    # a recurrence of hundreds of dependent steps with 512 units end to end: measured 0.9e-2 .. 1.1e-2 (2 x measured)
MATCHED_GRAD_WIDE8 = 3.5e-2     # the 8-convolution wide recipe end to end: measured 1.7e-2 (2 x measured)
MATCHED_LOSS_REL = 1e-3
# end-to-end bars of the convolutional recipes, (deep, weight-normalised) -> (logits, parameter gradients, weight-norm g gradients):
# 2 x what is measured.  The layer-by-layer tests below are the tight ones (5e-4 for every layer of every recipe, no accumulation).
END_TO_END_BARS = {(False, False): (2e-3, MATCHED_GRAD_REL_L2, 0.0), (False, True): (2e-3, MATCHED_GRAD_REL_L2, 3 * MATCHED_GRAD_REL_L2),
                   (True, False): (5e-3, MATCHED_GRAD_WIDE8, 0.0), (True, True): (3.5e-2, 0.3, 0.35)}
# The half build (ASR_ACT=f16; BARS measured there as well): logits agree BETTER (2e-4), but half's 8 x finer grid leaves 10 - 60 % of the
# activations one ulp apart between the two implementations (bfloat16: 2 - 40 %), and among near-ties of a Maxout / max-pooling pair that
# flips an arg-max now and then -- one gradient element routed to the other channel is 3e-3 .. 3e-2 of a small layer's gradient norm
# (tools/debug_f16_chain.py shows the error entering AT a maxout backward: 1.1e-4 above it, 2.3e-3 below).  zhang+residual/4 measures
# 6.9e-3 where the recipes without such an event measure 6e-4; the teacher-forced layer-by-layer test holds 5e-4 for every layer.
if BF16 is torch.float16:
    END_TO_END_BARS[(False, False)] = (2e-3, 1.5e-2, 0.0)
# (True, True): the 8-convolution wide branch WITH weight normalisation and its data-dependent initialisation is ill-conditioned end to
# end at random initialisation -- the rounding-matched oracle (logits 1.7e-2, gradients 0.15 - 0.17) tracks the device no better than
# the plain float32 oracle does (3e-2 / 0.34) once it forms its own W (which agrees with the device's to 5e-7, checked above): a
# last-bit difference of W flips bf16 roundings that eight normalised layers and four residual blocks amplify.  That recipe's
# correctness rests on the layer-by-layer test (every layer at 5e-4) and on the W comparison; the end-to-end bar is 2 x measured.



@pytest.mark.parametrize("B,T,bidir,V,H", [(3, 60, True, 29, 64), (4, 41, False, 32, 64), (4, 50, True, 32, 128)])
def test_forward_backward_matches_oracle(device, B, T, bidir, V, H):
    """logits, loss and every parameter gradient of one step: against the rounding-matched oracle at MATCHED_GRAD_REL_L2 relative
    L2 per parameter, and against the float32 oracle at bf16 tolerance.  V = 29: three-kernel logit region; V = 32: the fused
    LayerNorm + CTC backward (V % 4 == 0); H = 128: the partial-sum backward recurrence."""
    from asr import _ops
    from asr.loss import connectionist_temporal_classification
    cfg, model = _build(device, V=V, rnn=H, bidir=bidir)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=3, Lmax=8, seed=1, ragged=True)
    ys = model(x.to(device))
    assert isinstance(ys, tuple) and len(ys) == T and ys[0].shape == (B, V) and ys[0].dtype == torch.float32
    before = _ops.CALLS.get("layernorm_ctc_bwd", 0)
    loss = connectionist_temporal_classification(ys, labels.to(device), 0, x_len.to(device), l_len.to(device))
    loss.backward()
    from asr.functions import join_side_stream
    join_side_stream()
    torch.cuda.synchronize()
    fused = _ops.CALLS.get("layernorm_ctc_bwd", 0) > before
    assert fused == (V % 4 == 0)
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    logits = torch.stack(ys).detach().cpu()
    gi_bf16 = _ops.gru_gi_dtype(T, B, H, 2 if bidir else 1) == BF16
    report = {}
    for matched in (True, False):
        ref = omodel.DS2Oracle(state, cfg.num_conv_layers, cfg.num_rnn_layers, bidir, matched=matched, gi_bf16=gi_bf16, fused_logit_bias=fused,
                               gates_f16=_ops.gru_gates_f16(T, B, H, 2 if bidir else 1))
        logits_ref = ref(x)
        loss_ref = omodel.ctc_mean_loss(logits_ref, labels, x_len, l_len)
        loss_ref.backward()
        errs = {name: _rel(p.grad.cpu(), ref.g(name).grad) for name, p in model.named_parameters()}
        worst = max(errs, key=errs.get)
        report[matched] = (abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), _rel(logits, logits_ref.detach()), errs[worst], worst)
        if matched:
            assert report[True][0] < MATCHED_LOSS_REL, report[True]
            assert report[True][1] < 2e-3, report[True]
            for name, e in errs.items():
                assert e < MATCHED_GRAD_REL_L2, (name, e)
        else:
            assert _cos(logits, logits_ref.detach()) > 0.999
            assert report[False][0] < 2e-2
            for name, e in errs.items():
                assert e < 0.25, (name, e)
    print("model step B=%d T=%d V=%d H=%d: matched oracle loss %.1e logits %.1e worst grad %.1e (%s); float32 oracle loss %.1e logits %.1e "
          "worst grad %.1e (%s)" % ((B, T, V, H) + report[True] + report[False]))


def test_train_steps_follow_oracle(device):
    """three optimiser steps (clip 1, decay 1e-5, Adam 1e-3): losses and parameters stay with the CPU oracle's."""
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import Adam, GradientClipping, WeightDecay
    B, T, V = 4, 50, 29
    cfg, model = _build(device, V=V, seed=3)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=3, Lmax=8, seed=2)
    xd, ld, xl, ll = x.to(device), labels.to(device), x_len.to(device), l_len.to(device)
    model(xd)       # materialise lazily-sized parameters
    state = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref = omodel.DS2Oracle(state, cfg.num_conv_layers, cfg.num_rnn_layers, True)
    m = [torch.zeros_like(p) for p in ref.parameters()]
    v = [torch.zeros_like(p) for p in ref.parameters()]
    opt = Adam(alpha=1e-3, beta1=0.9)
    opt.setup(model)
    opt.add_hook(GradientClipping(1.0))
    opt.add_hook(WeightDecay(1e-5))
    for step in (1, 2, 3):
        loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
        opt.update(lossfun=lambda: loss)
        loss_ref, _ = omodel.train_step(ref, m, v, step, x, labels, x_len, l_len)
        assert abs(loss.item() - loss_ref) / abs(loss_ref) < 3e-2, (step, loss.item(), loss_ref)
    for name, p in model.named_parameters():
        d = (p.detach().cpu() - ref.g(name).detach()).abs()
        # three Adam steps move each weight by <= 3e-3; a gradient whose sign differs (|g| ~ 0 in bf16) costs 2e-3/step
        assert d.max().item() < 7e-3 and d.mean().item() < 6e-4, (name, d.max().item(), d.mean().item())


def test_reference_style_stream_model(device):
    """A 'zhang'-style nn.Stream stack exactly as run/ctc/cnn/model.py:40-89 writes it (explicit pad + slice lambda,
    Maxout, MaxPooling2D, kernel_height conv, 1x1 convs, LayerNormalization) runs and back-propagates."""
    import math
    import asr.nn as nn
    from asr.model.cnn import AcousticModel
    from asr.loss import connectionist_temporal_classification
    torch.manual_seed(0)
    ndim_h, ndim_dense, V, nmel = 16, 24, 13, 40
    ks, pad = (3, 5), 4
    kernel_height = int(math.ceil((nmel - 2) / 3))
    model = AcousticModel()
    model.layer(nn.Convolution2D(3, ndim_h * 2, ks, stride=1, pad=(0, pad)), lambda x: x[..., :-pad], nn.Maxout(2),
                nn.Dropout(0), nn.MaxPooling2D(ksize=(3, 1)))
    model.layer(nn.Residual(nn.Convolution2D(ndim_h, ndim_h * 2, ks, stride=1, pad=(1, pad)), lambda x: x[..., :-pad],
                            nn.Maxout(2), nn.Dropout(0)))
    model.layer(nn.Convolution2D(ndim_h, ndim_dense * 2, ksize=(kernel_height, 1), stride=1, pad=0), nn.Maxout(2), nn.Dropout(0))
    model.layer(nn.Convolution2D(ndim_dense, ndim_dense * 2, ksize=1, stride=1, pad=0), nn.Maxout(2), nn.Dropout(0))
    model.layer(nn.Convolution2D(ndim_dense, V, ksize=1, stride=1, pad=0), nn.LayerNormalization(None))
    model.to_gpu()
    B, T = 3, 30
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=2, Lmax=5, seed=4)
    ys = model(x.to(device))
    assert len(ys) == T and ys[0].shape == (B, V)
    loss = connectionist_temporal_classification(ys, labels.to(device), 0, x_len.to(device), l_len.to(device))
    loss.backward()
    assert np.isfinite(loss.item())
    for name, p in model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), name
    y2 = model(x.to(device), split_into_variables=False)
    assert y2.shape == (B, T, V)
    # torch-CPU restatement of the same stack
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    from oracle import nn as onn
    h = onn.maxpool_h(onn.maxout2(onn.conv2d_causal(x, sd["layer_0.W"], sd["layer_0.b"], 0)), 3)
    h = onn.maxout2(onn.conv2d_causal(h, sd["layer_5_0.W"], sd["layer_5_0.b"], 1)) + h
    h = onn.maxout2(torch.nn.functional.conv2d(h, sd["layer_6.W"], sd["layer_6.b"]))
    h = onn.maxout2(torch.nn.functional.conv2d(h, sd["layer_9.W"], sd["layer_9.b"]))
    h = torch.nn.functional.conv2d(h, sd["layer_12.W"], sd["layer_12.b"])
    yr, _ = onn.layer_normalization(h.numpy().astype(np.float64), sd["layer_13.gamma"].numpy().astype(np.float64),
                                    sd["layer_13.beta"].numpy().astype(np.float64))
    got = y2.detach().cpu().numpy()                      # (B, T, V)
    want = yr[:, :, 0, :].transpose(0, 2, 1)
    assert _cos(torch.tensor(got), torch.tensor(want)) > 0.999


@pytest.mark.parametrize("arch", ["zhang", "zhang+fc_relu", "zhang+residual", "zhang+layernorm", "glu", "relu+layernorm",
                                  "relu+layernorm+residual"])
def test_cnn_recipes_train_step(device, arch):
    """every recipe of run/ctc/cnn/model.py runs forward + CTC + backward + one optimiser step and lowers its loss"""
    from asr.model import cnn
    from asr.model.architectures import build_model
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import get_optimizer, GradientClipping, WeightDecay
    torch.manual_seed(0)
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers, cfg.architecture = 19, 3, 16, 24, 3, arch
    model = build_model(cfg).to_gpu()
    x, labels, x_len, l_len = omodel.synthetic_batch(3, 40, 19, Lmin=2, Lmax=6, seed=5)
    xd, ld, xl, ll = x.to(device), labels.to(device), x_len.to(device), l_len.to(device)
    opt = get_optimizer("adam", 2e-3, 0.9)
    ys = model(xd)
    opt.setup(model)
    opt.add_hook(GradientClipping(1.0))
    opt.add_hook(WeightDecay(1e-5))
    if LOSS_SCALE != 1.0:
        opt.loss_scaling()          # dynamic, chainer.Optimizer.loss_scaling
    losses = []
    # half build: a recipe whose gradients times the initial scale of 4096 leave the half range drops those steps, halving the scale each
    # time, until the backward pass fits (the un-normalised `glu` recipe at this toy size needs S < 1: its activation gradients reach 4e4)
    for _ in range(4 if LOSS_SCALE == 1.0 else 28):
        loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
        opt.update(lossfun=lambda: loss)
        losses.append(loss.item())
        if opt.applied_steps() == 4:
            break
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert opt.applied_steps() == 4 and opt.applied_steps() + opt.loss_scale()[1] == len(losses)


def test_weight_copies_refreshed_in_one_launch(device):
    """after optimizer.update every registered bf16 weight copy (plain / transposed / per-direction transposed) equals
    the cast of the NEW master weights, is the same tensor as before (refreshed in place by asr_cast_bf16_many) and is
    found current by the next forward pass (no per-layer cast launches)"""
    from asr import link as L
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import get_optimizer, GradientClipping
    cfg, model = _build(device, V=29)
    x, labels, x_len, l_len = omodel.synthetic_batch(3, 40, 29, Lmin=3, Lmax=8, seed=5)
    args = (labels.to(device), 0, x_len.to(device), l_len.to(device))
    opt = get_optimizer("adam", 1e-2, 0.9)
    model(x.to(device))
    opt.setup(model)
    opt.add_hook(GradientClipping(1.0))
    # 1st update: the optimiser moves the parameters into its flat buffer (copies made before that are dropped);
    # 2nd update: copies made lazily from the flat buffer and registered
    for _ in range(2):
        opt.update(lossfun=lambda: connectionist_temporal_classification(model(x.to(device)), *args))
    before = {(id(m), k): v[1] for m in model.modules() for k, v in getattr(m, "_cast_registry", {}).items()}
    assert len(before) >= 6
    masters = {(id(m), k): v[0].detach().clone() for m in model.modules() for k, v in getattr(m, "_cast_registry", {}).items()}
    opt.update(lossfun=lambda: connectionist_temporal_classification(model(x.to(device)), *args))
    assert getattr(model, "_cast_table", None) is not None and model._cast_table[2] >= len(before)
    seen = 0
    for m in model.modules():
        for key, (param, out, jobs) in getattr(m, "_cast_registry", {}).items():
            assert out is before[(id(m), key)]                         # refreshed in place
            assert not torch.equal(param.detach(), masters[(id(m), key)])    # the step did move the weights
            stamp, cached = m._compute_cache[key][:2]
            assert cached is out and stamp[0] == L._WEIGHT_EPOCH[0]    # current for the next forward pass
            w = param.detach()
            for so, do, r, c, t in jobs:
                src = w.reshape(-1)[so:so + r * c].reshape(r, c)
                want = (src.t() if t else src).contiguous().to(BF16)
                got = out.reshape(-1)[do:do + r * c].reshape(want.shape)
                assert torch.equal(got, want), (key, r, c, t)
                seen += 1
    assert seen >= len(before)


def test_cast_many_ragged_shapes(device):
    """asr_cast_bf16_many on shapes that are not multiples of the 64x64 tile or of 4"""
    from asr import _lib
    torch.manual_seed(3)
    shapes = [(1, 1, 0), (5, 7, 1), (64, 64, 1), (65, 130, 0), (130, 65, 1), (3, 257, 1), (200, 4, 1), (96, 1536, 0), (7, 12, 1)]
    srcs = [torch.randn(r, c, device=device) for r, c, _ in shapes]
    dsts = [torch.full((c, r) if t else (r, c), 7.0, dtype=BF16, device=device) for r, c, t in shapes]
    rows, tile = [], 0
    for (r, c, t), s, d in zip(shapes, srcs, dsts):
        rows.append([s.data_ptr(), d.data_ptr(), r, c, t, tile])
        tile += ((r + 63) // 64) * ((c + 63) // 64)
    table = torch.tensor(rows, dtype=torch.int64).to(device)
    _lib.check(_lib.lib().asr_cast_bf16_many(_lib.stream(), _lib.ptr(table), len(rows), tile), "asr_cast_bf16_many")
    torch.cuda.synchronize()
    for (r, c, t), s, d in zip(shapes, srcs, dsts):
        want = (s.t() if t else s).contiguous().to(BF16)
        assert torch.equal(d, want), (r, c, t)


def test_step_is_dropped_on_the_device_when_a_recurrence_gave_up(device):
    """a persistent GRU launch that abandons an in-launch wait leaves finite garbage behind, which the
    isfinite(norm) guard lets through.  asr_step_control drops the step on the device when an abort word is raised: the
    parameters, Adam's moments and the applied-step count stay untouched; the host learns one step later (AsrHipError) and
    training resumes after that."""
    from asr import _ops, _lib
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import Adam, GradientClipping, WeightDecay
    B, T, V = 4, 50, 29
    cfg, model = _build(device, V=V, seed=3)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=3, Lmax=8, seed=2)
    xd, ld, xl, ll = x.to(device), labels.to(device), x_len.to(device), l_len.to(device)
    opt = Adam(alpha=1e-3, beta1=0.9)
    model(xd)
    opt.setup(model)
    opt.add_hook(GradientClipping(1.0))
    opt.add_hook(WeightDecay(1e-5))

    def step():
        loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
        opt.update(lossfun=lambda: loss)
    step()
    torch.cuda.synchronize()
    assert opt.applied_steps() == 1
    snap = [t.clone() for t in (opt.flat_parameters(), opt.m, opt.v)]
    _ops.LAST_SYNC[0][1023:1024].fill_(1)             # forge: "a launch gave up"
    step()                                            # every recurrence of this step gives up at once: garbage gradients
    torch.cuda.synchronize()
    for a, b_ in zip(snap, (opt.flat_parameters(), opt.m, opt.v)):
        assert torch.equal(a, b_)
    assert opt.applied_steps() == 1 and float(opt._flat["ctl"][0].item()) == 1.0
    with pytest.raises(_lib.AsrHipError):             # the host notices one step later, without ever having synchronised
        step()
    torch.cuda.synchronize()
    for a, b_ in zip(snap, (opt.flat_parameters(), opt.m, opt.v)):
        assert torch.equal(a, b_)
    step()                                            # the word was cleared: back to normal
    torch.cuda.synchronize()
    assert opt.applied_steps() == 2 and not torch.equal(snap[0], opt.flat_parameters())
    # the evaluation path checks too (it synchronises anyway)
    from asr import error
    _ops.LAST_SYNC[0][1023:1024].fill_(1)
    ids = torch.zeros(2, 5, dtype=torch.int32, device=device)
    with pytest.raises(_lib.AsrHipError):
        error.compute_minibatch_error(ids, ids, 0, None, None)
    _ops.gru_check_all()                              # cleared


def test_optimiser_state_survives_a_reflatten(device):
    """attaching / detaching a communicator or a parameter appearing late rebuilds the flat buffers; Adam's
    moments and the applied-step count must come along"""
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import Adam
    B, T, V = 3, 40, 29
    cfg, model = _build(device, V=V, seed=5)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=3, Lmax=8, seed=2)
    xd, ld, xl, ll = x.to(device), labels.to(device), x_len.to(device), l_len.to(device)
    opt = Adam(alpha=1e-3, beta1=0.9)
    model(xd)
    opt.setup(model)
    for _ in range(2):
        loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
        opt.update(lossfun=lambda: loss)
    torch.cuda.synchronize()
    params = list(opt._params())
    m_before = {id(p): opt.m[o:o + p.numel()].clone() for p, o in zip(params, opt._flat["offsets"])}
    w_before = {id(p): p.detach().clone() for p in params}
    # force a re-flatten: one parameter moves to fresh storage (what a lazily (re)initialised layer does)
    p0 = params[3]
    p0.data = p0.data.clone()
    opt._ensure_flat()
    assert opt.applied_steps() == 2
    for p, o in zip(opt._params(), opt._flat["offsets"]):
        assert torch.equal(opt.m[o:o + p.numel()], m_before[id(p)])
        assert torch.equal(p.detach(), w_before[id(p)])
    assert float(opt.m.abs().sum()) > 0


def test_save_fresh_build_load_gives_identical_logits(device, tmp_path):
    """row f4 + ADVICE r1: model.save(path); build_model(config); model.load(path) on the FRESH model (lazily sized
    parameters still empty) reproduces the logits bit for bit; the file carries the reference's parameter paths"""
    import numpy as np
    from asr import serializers
    cfg, model = _build(device, V=29, seed=11)
    x, *_ = omodel.synthetic_batch(3, 40, 29, Lmin=3, Lmax=8, seed=5)
    xd = x.to(device)
    y0 = torch.stack(model(xd)).detach().clone()
    path = str(tmp_path / "model.hdf5")
    model.save(path)
    keys = set(np.load(path).files) if serializers.sniff(path) == "npz" else None
    if keys is not None:
        assert "_module_rnn_blocks_sequential_0/w_ih" in keys and "_module_dense_blocks_sequential_7/norm/gamma" in keys
    cfg2, fresh = _build(device, V=29, seed=12)
    assert fresh.load(path) is True
    y1 = torch.stack(fresh(xd)).detach()
    assert torch.equal(y0, y1)


def test_joint_gram_ctc_and_ctc_on_one_model_output(device):
    """run/gram_ctc/cnn/train.py:163-167,195-198 (--joint-training): loss = gram_ctc(y, t, bigram, ...) + CTC(y, t, ...) on
    the SAME logits.  Two consumers of the float32 logits: each loss hands autograd a float32 gradient, LayerNormalization's
    backward sums them and posts ONE bf16 gradient into the projection's mailbox.  Checked: the joint loss is the sum of the
    two, and every parameter gradient of the joint step equals the sum of the gradients of the two separate steps."""
    from asr.loss import connectionist_temporal_classification, gram_ctc
    from asr.data.synthetic import synthetic_batch, synthetic_gram_labels
    from oracle import ctc as octc
    B, T, V, nuni = 4, 60, 60, 20
    x, labels, x_len, l_len = synthetic_batch(B, T, nuni, Lmin=3, Lmax=8, seed=3, ragged=True)
    bigram = synthetic_gram_labels(labels, l_len, V, first_bigram=nuni, seed=3)
    xd, ld, bd, xl, ll = (t.to(device) for t in (x, labels, bigram, x_len, l_len))

    def run(which):
        cfg, model = _build(device, V=V, seed=21)
        ys = model(xd)
        terms = []
        if which in ("gram", "joint"):
            terms.append(gram_ctc(ys, ld, bd, 0, xl, ll))
        if which in ("ctc", "joint"):
            terms.append(connectionist_temporal_classification(ys, ld, 0, xl, ll))
        loss = terms[0] if len(terms) == 1 else terms[0] + terms[1]
        loss.backward()
        from asr.functions import join_side_stream
        join_side_stream()
        torch.cuda.synchronize()
        return loss.item(), torch.stack(tuple(ys)).detach().cpu(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}

    lg, logits, gg = run("gram")
    lc, _, gc = run("ctc")
    lj, _, gj = run("joint")
    assert abs(lj - (lg + lc)) <= 1e-5 * abs(lj)
    # both loss values against the float64 oracle on the device's own logits (1e-4: BASELINE.json)
    lo_g, _ = octc.gram_ctc_loss_grad(logits.numpy(), labels.numpy(), bigram.numpy(), 0, x_len.numpy(), l_len.numpy(), "mean")
    lo_c, _ = octc.ctc_loss_grad(logits.numpy(), labels.numpy(), 0, x_len.numpy(), l_len.numpy(), "mean")
    assert abs(lg - lo_g) <= 1e-4 * abs(lo_g) and abs(lc - lo_c) <= 1e-4 * abs(lo_c)
    for name in gj:
        want = gg[name] + gc[name]
        err = float((gj[name] - want).norm() / (want.norm() + 1e-30))
        assert err < 2e-2, (name, err)          # one bf16 rounding of the summed logit gradient instead of two


@pytest.mark.parametrize("arch,nconv,wn", [("zhang", 3, False), ("zhang+fc_relu", 2, False), ("zhang+residual", 4, False),
                                           ("zhang+residual", 6, False), ("zhang+residual", 5, True), ("zhang+layernorm", 2, False),
                                           ("glu", 2, False), ("glu", 2, True), ("relu+layernorm", 2, False),
                                           ("relu+layernorm+residual", 3, False)])
def test_cnn_recipes_match_the_oracle_end_to_end(device, arch, nconv, wn):
    """row a19 / BASELINE configs[4]: every recipe of run/ctc/cnn/model.py -- incl. `zhang+residual` at 4 conv layers and its
    wide "VGG-deep" branch (num_conv_layers > 4, :153-157,177-187) and the weight-normalised variants -- forward + CTC +
    backward on the HIP path against oracle/cnn.py (torch-CPU fp32 restatement of the recipe) with the same parameters:
    logits, loss and every parameter gradient."""
    from asr.model import cnn
    from asr.model.architectures import build_model
    from asr.loss import connectionist_temporal_classification
    from oracle import cnn as ocnn
    torch.manual_seed(3)
    V, B, T = 19, 3, 36
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers = V, 3, 16, 24, nconv
    cfg.architecture, cfg.weightnorm = arch, wn
    model = build_model(cfg).to_gpu()
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=2, Lmax=6, seed=7, ragged=True)
    xd = x.to(device)
    with torch.no_grad():
        model(xd)                                   # lazily sized parameters; data-dependent weight-norm initialisation
    ys = model(xd)
    loss = connectionist_temporal_classification(ys, labels.to(device), 0, x_len.to(device), l_len.to(device))
    # half build: the un-normalised `glu` recipe at this toy size has activation gradients of 4e4 -- beyond the half range WITHOUT any
    # scale (the dynamic loss scale settles below 1 for it: test_cnn_recipes_train_step); its seed is 2^-6
    seed = LOSS_SCALE if not (LOSS_SCALE != 1.0 and arch == "glu" and not wn) else 2.0 ** -6
    _seeded_backward(loss, seed)
    from asr.functions import join_side_stream
    join_side_stream()
    torch.cuda.synchronize()
    logits = torch.stack(tuple(ys)).detach().float().cpu()
    report = {}
    # A weight-normalised W = g V / (||V|| + 1e-9) (asr/nn/convolution_2d.py:21-25,62-64; the formula is pinned by tests/golden/norm.npz):
    # the device's W is compared with the oracle's HERE, in float32, and the end-to-end run below uses the ORACLE's own W (VERDICT r3
    # next 4c: the oracle is no longer handed the device's value).  The two differ in the last float32 bit (order of the norm's sum),
    # which flips a few bf16 roundings of W: ~3e-4 per weight-normalised layer on the end-to-end figures, inside the bars below.
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    nwn = 0
    for name, mod in model.named_modules():
        if hasattr(mod, "V") and hasattr(mod, "g") and mod.g.numel() > 0:
            V_, g_ = sd[name + ".V"], sd[name + ".g"]
            W_ref = g_ * V_ / (torch.sqrt((V_ * V_).sum(dim=(1, 2, 3), keepdim=True)) + 1e-9)
            w_err = _rel(mod.W.detach().float().cpu(), W_ref)
            assert w_err < 5e-7, (name, w_err)
            nwn += 1
    assert (nwn > 0) == bool(wn)
    # two correct bf16 implementations drift apart with depth (a flipped rounding is amplified by the layers behind it: see the
    # layer-by-layer test below, which is the tight one); the 8-convolution wide branch gets the wider end-to-end bar
    deep = nconv > 4
    for matched in (True, False):
        params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
        out = ocnn.forward(arch, cfg, params, x, matched=matched, fused_logit_bias=False)        # V = 19: three-kernel logit region
        assert out.shape == (B, V, 1, T)
        logits_ref = ocnn.logits_tbv(out)
        loss_ref = omodel.ctc_mean_loss(logits_ref, labels, x_len, l_len)
        (loss_ref * seed).backward()       # both sides' gradients carry the seed: the relative errors do not
        errs = {}
        for name, p in model.named_parameters():
            g_ref = params[name].grad
            assert p.grad is not None and g_ref is not None, name
            errs[name] = _rel(p.grad.cpu(), g_ref)
        worst = max(errs, key=errs.get)
        report[matched] = (abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), _rel(logits, logits_ref.detach()), errs[worst], worst)
        if matched:         # one bar for every parameter, large or small (the float32 oracle needed 0.90 cosine for 16-element ones)
            worst_g = max([e for n, e in errs.items() if n.endswith(".g")] or [0.0])
            worst_other = max(e for n, e in errs.items() if not n.endswith(".g"))
            print("   matched, by class: loss %.1e logits %.1e worst .g gradient %.1e worst other gradient %.1e" % (report[True][0], report[True][1], worst_g, worst_other))
            assert report[True][0] < MATCHED_LOSS_REL, report[True]
            assert report[True][1] < END_TO_END_BARS[(deep, bool(wn))][0], report[True]
            for name, e in errs.items():
                assert e < END_TO_END_BARS[(deep, bool(wn))][2 if name.endswith(".g") else 1], (name, e)
            # the wide weight-normalised recipe's 0.3 bar alone would let a MISSING term through (half of a bias gradient dropped
            # is a relative error of ~0.3 - 0.5 with a norm ratio of ~0.7): direction and size of every gradient are held as well --
            # rounding noise of the measured 0.15 - 0.17 leaves the cosine above 0.98 and the norms within a few per cent
            for name, p in model.named_parameters():
                g_dev, g_ref = p.grad.detach().float().cpu().flatten().double(), params[name].grad.flatten().double()
                ratio = float(g_dev.norm() / (g_ref.norm() + 1e-30))
                cosine = float(g_dev @ g_ref / (g_dev.norm() * g_ref.norm() + 1e-30))
                assert cosine > 0.93 and 0.8 < ratio < 1.25, (name, cosine, ratio)
        else:
            assert _cos(logits, logits_ref.detach()) > 0.998
            assert report[False][0] <= 3e-2
            for name, e in errs.items():
                assert e < 0.45, (name, e)
    print("recipe %s/%d/wn=%s: matched oracle loss %.1e logits %.1e worst grad %.1e (%s); float32 oracle loss %.1e logits %.1e worst grad "
          "%.1e (%s)" % ((arch, nconv, wn) + report[True] + report[False]))


def test_fused_logit_region_gives_the_gradients_of_the_unfused_one(device):
    """model level: with the CTC gradient formed inside LayerNormalization's backward (and the projection's bias gradient taken
    from the same sweep) every parameter gradient equals the one of the three-kernel route up to float32 summation order /
    one bf16 rounding of dx"""
    from asr import functions as F, _ops
    from asr.loss import connectionist_temporal_classification
    B, T, V = 4, 48, 40
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=3, Lmax=8, seed=9, ragged=True)
    xd, ld, xl, ll = x.to(device), labels.to(device), x_len.to(device), l_len.to(device)

    def run(fused):
        F.FUSE_CTC_INTO_LAYERNORM[0] = fused
        try:
            cfg, model = _build(device, V=V, seed=31)
            before = _ops.CALLS.get("layernorm_ctc_bwd", 0)
            loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
            loss.backward()
            F.join_side_stream()
            torch.cuda.synchronize()
            assert (_ops.CALLS.get("layernorm_ctc_bwd", 0) - before) == (1 if fused else 0)
            return loss.item(), {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()}
        finally:
            F.FUSE_CTC_INTO_LAYERNORM[0] = True
    (lf, gf), (lu, gu) = run(True), run(False)
    assert lf == lu
    for name in gu:
        err = float((gf[name] - gu[name]).norm() / (gu[name].norm() + 1e-30))
        assert err < 1e-2, (name, err)
    # the logit projection's bias gradient comes out of the fused sweep in float32 (no bf16 rounding of dx in between)
    name = "dense_blocks._sequential_6.b"
    assert float((gf[name] - gu[name]).norm() / gu[name].norm()) < 5e-3


@pytest.mark.parametrize("kind", ["ds2", "cnn", "cnn+weightnorm"])
def test_fused_first_block_gives_the_model_the_gradients_of_the_three_passes(device, kind):
    """model level: the first convolution + Maxout + MaxPooling as one pass (csrc/conv_first.hip, the default where the layer has a
    multiple of 128 output channels) against the same model with ASR_DEBUG conv_mp=0 semantics: same logits, same loss, every parameter
    gradient equal up to float32 summation order -- for the conv + recurrent model, a convolutional recipe and its weight-normalised form
    (whose first convolution hands the gradient of a derived weight back to the tape)"""
    if BF16 is torch.float16 and kind == "ds2":
        pytest.skip("the recurrences are bfloat16-only")
    from asr import functions as F, _ops
    from asr.loss import connectionist_temporal_classification
    B, T, V = 3, 40, 23
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=2, Lmax=6, seed=4, ragged=True)
    xd, ld, xl, ll = x.to(device), labels.to(device), x_len.to(device), l_len.to(device)

    def build():
        torch.manual_seed(17)
        if kind == "ds2":
            from asr.model import ds2
            cfg = ds2.configure()
            cfg.vocab_size, cfg.ndim_conv, cfg.ndim_rnn, cfg.ndim_dense, cfg.num_rnn_layers = V, 64, 64, 32, 1
            return ds2.Model(cfg).to_gpu()
        from asr.model import cnn
        from asr.model.architectures import build_model
        cfg = cnn.configure()
        cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers = V, 3, 64, 24, 2
        cfg.architecture, cfg.weightnorm = "zhang", kind.endswith("weightnorm")
        return build_model(cfg).to_gpu()

    def run(fused):
        F.CONV_MP[0] = fused
        try:
            model = build()
            with torch.no_grad():
                model(xd)                   # lazily sized parameters; data-dependent weight-norm initialisation (never fused)
            before = _ops.CALLS.get("conv_mp_fwd", 0)
            ys = model(xd)
            loss = connectionist_temporal_classification(ys, ld, 0, xl, ll)
            _seeded_backward(loss)
            F.join_side_stream()
            torch.cuda.synchronize()
            assert (_ops.CALLS.get("conv_mp_fwd", 0) - before) == (1 if fused else 0)
            return (loss.item(), torch.stack(tuple(ys)).detach().float().cpu(),
                    {n: p.grad.detach().float().cpu().clone() for n, p in model.named_parameters()})
        finally:
            F.CONV_MP[0] = True
    (lf, yf, gf), (lu, yu, gu) = run(True), run(False)
    assert torch.equal(yf, yu) and lf == lu
    assert set(gf) == set(gu)
    for name in gu:
        err = float((gf[name] - gu[name]).norm() / (gu[name].norm() + 1e-30))
        assert err < 1e-4, (name, err)


def test_abort_word_of_a_third_stream_is_seen(device):
    """one control buffer per (device, stream) that launched a recurrence; a process that ran recurrences on three
    streams has three abort words, and the step control used to look at the first two.  All of them are ORed on the device now
    (asr_gather_abort): raise the LAST one and the step must be dropped."""
    from asr import _ops, _lib
    from asr.loss import connectionist_temporal_classification
    from asr.optimizers import Adam, GradientClipping
    B, T, V = 4, 40, 29
    cfg, model = _build(device, V=V, seed=3)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=3, Lmax=8, seed=2)
    xd, ld, xl, ll = x.to(device), labels.to(device), x_len.to(device), l_len.to(device)
    opt = Adam(alpha=1e-3, beta1=0.9)
    model(xd)
    opt.setup(model)
    opt.add_hook(GradientClipping(1.0))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    with torch.no_grad():
        for st in streams:                      # forward passes on two more streams: two more control buffers
            st.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(st):
                model(xd)
            torch.cuda.current_stream().wait_stream(st)
    torch.cuda.synchronize()
    words = [w for w in _ops.abort_words() if w.device == device]
    assert len(words) >= 3

    def step():
        loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
        opt.update(lossfun=lambda: loss)
    step()
    torch.cuda.synchronize()
    assert opt.applied_steps() == 1
    snap = opt.flat_parameters().clone()
    last = list(_ops._SYNC.keys())[-1]
    assert last[1] != torch.cuda.current_stream().cuda_stream          # not the stream the train step launches on
    _ops._SYNC[last][1023:1024].fill_(1)
    step()
    torch.cuda.synchronize()
    assert torch.equal(snap, opt.flat_parameters()) and opt.applied_steps() == 1
    assert float(opt._flat["ctl"][5].item()) == 1.0
    with pytest.raises(_lib.AsrHipError):
        step()
    torch.cuda.synchronize()
    step()                                      # every word was cleared when the host was told
    torch.cuda.synchronize()
    assert opt.applied_steps() == 2


@pytest.mark.parametrize("bidir", [False, True])
def test_configs0_literal_shape(device, bidir):
    """BASELINE configs[0] at its literal shape on the HIP path: 2 x conv + ONE GRU layer (ndim_conv 64, 512 units, dense 320), B=4,
    T=200, V=119 (the reference's unigram inventory), one forward + CTC + backward against the float32 oracle (the configuration's
    "Chainer CPU reference" stand-in) and the rounding-matched oracle"""
    from asr import _ops
    from asr.loss import connectionist_temporal_classification
    B, T, V = 4, 200, 119
    cfg, model = _build(device, V=V, conv=64, rnn=512, dense=320, nrnn=1, bidir=bidir, seed=9)
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=10, Lmax=30, seed=4, ragged=True)
    ys = model(x.to(device))
    loss = connectionist_temporal_classification(ys, labels.to(device), 0, x_len.to(device), l_len.to(device))
    loss.backward()
    from asr.functions import join_side_stream
    join_side_stream()
    torch.cuda.synchronize()
    _ops.gru_check_sync()
    state = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    gi_bf16 = _ops.gru_gi_dtype(T, B, 512, 2 if bidir else 1) == BF16
    for matched in (True, False):
        ref = omodel.DS2Oracle(state, cfg.num_conv_layers, 1, bidir, matched=matched, gi_bf16=gi_bf16, fused_logit_bias=False,
                               gates_f16=_ops.gru_gates_f16(T, B, 512, 2 if bidir else 1))
        loss_ref = omodel.ctc_mean_loss(ref(x), labels, x_len, l_len)
        loss_ref.backward()
        errs = {name: _rel(p.grad.cpu(), ref.g(name).grad) for name, p in model.named_parameters()}
        worst = max(errs, key=errs.get)
        lrel = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
        print("configs[0] bidir=%s matched=%s: loss %.2e worst gradient %.2e (%s)" % (bidir, matched, lrel, errs[worst], worst))
        assert lrel < (1e-4 if matched else 2e-2)
        assert errs[worst] < (MATCHED_GRAD_RNN512 if matched else 0.25), (worst, errs[worst])


# ---------------------------------------------------------------------------------------------- layer by layer
# End to end, two correct bf16 implementations drift apart with depth: a last-bit difference in a float32 sum flips one bf16 rounding,
# the next layers amplify it (measured on the 8-layer wide recipe: 6e-5 after the fourth convolution, 1.1e-3 four residual blocks
# later, 1.7e-2 in the first layer's gradient -- with every layer correct).  So the end-to-end gate above cannot be both tight and
# depth independent.  This test is: every layer (a Residual block counts as one, Maxout + MaxPooling as the fused pair the device
# runs) is given the DEVICE's own input and the DEVICE's own output gradient, and its output, its input gradient and its parameter
# gradients must match the rounding-matched oracle of that ONE layer -- no accumulation, one bar for every layer of every recipe.
LAYER_REL_L2 = 5e-4
LAYER_GX_COL2IM = 5e-3


def _trace_layers(model, xd):
    """model(xd) with every top-level layer's input / output recorded and a hook on every output's gradient"""
    from asr.nn import nn as nnmod
    from asr import functions as F
    rec = []
    orig = nnmod._apply_layers

    def traced(layers, x):
        if layers is not model.layers:
            return orig(layers, x)
        i = 0
        while i < len(layers):
            j = nnmod._fusable_pool(layers, i) if x.dim() == 4 else -1
            xin = x
            if j > 0:
                ks = layers[j].ksize
                x = F.maxout_max_pooling(x, ks[0] if isinstance(ks, (tuple, list)) else ks, sole_consumer=i > 0)
                span = (i, j + 1)
                i = j + 1
            else:
                y = layers[i](x)
                if isinstance(layers[i], nnmod.Residual):
                    y = F.add(y, x)
                x = y
                span = (i, i + 1)
                i += 1
            entry = {"span": span, "xin": xin.detach(), "xout": x.detach(), "gout": None}
            if x.requires_grad:
                x.register_hook(lambda g, e=entry: e.__setitem__("gout", g.detach().clone()))
            rec.append(entry)
        return x
    nnmod._apply_layers = traced
    try:
        ys = model(xd)
    finally:
        nnmod._apply_layers = orig
    return ys, rec


@pytest.mark.parametrize("arch,nconv,wn", [("zhang", 3, False), ("zhang", 3, True), ("zhang+fc_relu", 2, False), ("zhang+residual", 4, False),
                                           ("zhang+residual", 6, False), ("zhang+residual", 5, True), ("zhang+layernorm", 2, False),
                                           ("glu", 2, True), ("relu+layernorm", 2, False), ("relu+layernorm+residual", 3, False)])
def test_cnn_recipes_layer_by_layer_against_the_matched_oracle(device, arch, nconv, wn):
    from asr.model import cnn
    from asr.model.architectures import build_model
    from asr.loss import connectionist_temporal_classification
    from asr import functions as F
    from oracle import cnn as ocnn
    torch.manual_seed(3)
    V, B, T = 19, 3, 36
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers = V, 3, 16, 24, nconv
    cfg.architecture, cfg.weightnorm = arch, wn
    model = build_model(cfg).to_gpu()
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=2, Lmax=6, seed=7, ragged=True)
    xd = x.to(device)
    with torch.no_grad():
        model(xd)
    ys, rec = _trace_layers(model, xd)
    loss = connectionist_temporal_classification(ys, labels.to(device), 0, x_len.to(device), l_len.to(device))
    _seeded_backward(loss)         # teacher-forced below: the oracle's layers are handed the device's (scaled) gradients
    F.join_side_stream()
    torch.cuda.synchronize()
    prog = ocnn.program(arch, cfg)
    segs = ocnn.segments(prog)
    assert len(segs) == len(model.layers)
    # the VALUE of a weight-normalised W as the device formed it (see oracle.cnn.forward): the layers' W property re-runs the kernel
    weights = {}
    for name, mod in model.named_modules():
        if hasattr(mod, "V") and hasattr(mod, "g") and getattr(mod, "g").numel() > 0:
            weights[name] = mod.W.detach().float().cpu()
    dev_grads = {n: p.grad.detach().float().cpu() for n, p in model.named_parameters()}
    worst = {"y": (-1.0, ""), "gx": (-1.0, ""), "gp": (-1.0, "")}

    def note(kind, err, what):
        if err > worst[kind][0]:
            worst[kind] = (err, what)
    checked_grads = set()
    for k, e in enumerate(rec):
        lo, hi = segs[e["span"][0]][0], segs[e["span"][1] - 1][1]
        names = sorted({n for (_, n, _) in prog[lo:hi] if n is not None})
        params = {pn: v.detach().cpu().clone().requires_grad_(True) for pn, v in model.named_parameters() if pn.rsplit(".", 1)[0] in names}
        xin = e["xin"].float().cpu()
        if k == 0:
            xin = xin.to(BF16).float()           # the first layer packs the loader's float32 minibatch to bf16
        xin.requires_grad_(k > 0)
        y = ocnn.run(prog, lo, hi, params, xin, matched=True, fused_logit_bias=False, weights=weights)
        got = e["xout"].float().cpu()
        tag = "%s[%d:%d] %s" % (arch, lo, hi, ",".join(names))
        err = _rel(got, y.detach())
        note("y", err, tag)
        assert err < LAYER_REL_L2, ("forward", tag, err)
        gout = e["gout"]
        if gout is None or float(gout.float().abs().max()) == 0.0:
            continue            # the float32 logit layers hand their gradient over outside autograd (functions._GradMailbox): see test_ctc_gpu
        y.backward(gout.float().cpu())
        # a convolution whose Cout * kh * kw is not a multiple of 32 takes the column-matrix route backward (gemm + col2im): its
        # per-tap products are rounded to bf16 before col2im adds them -- one more rounding than the implicit kernel (every recipe
        # size of the reference takes the implicit one: Cout in {128, 256, 512, 640}; the 16-channel layers of this test do not).
        # Whatever lies upstream of such a convolution INSIDE the unit (the LayerNormalization of a pre-activation Residual) sees it too.
        def own(n):
            return params[n + (".V" if n + ".V" in params else ".W")]
        col2im = [n for (op, n, _) in prog[lo:hi] if op in ("conv", "glu") and tuple(own(n).shape[2:]) == (3, 5) and (own(n).shape[0] * 15) % 32 != 0]
        if k > 0 and rec[k - 1]["gout"] is not None and float(rec[k - 1]["gout"].float().abs().max()) > 0.0:
            # (the device's tensor is bf16: where two branches meet -- a Residual's skip and its convolution -- the sum is rounded once more)
            err = _rel(rec[k - 1]["gout"].float().cpu(), xin.grad.to(BF16).float())
            note("gx", err, tag)
            assert err < (LAYER_GX_COL2IM if col2im else LAYER_REL_L2), ("input gradient", tag, err)
        for pn, v in params.items():
            err = _rel(dev_grads[pn], v.grad)
            note("gp", err, pn)
            checked_grads.add(pn)
            upstream_of_col2im = bool(col2im) and pn.rsplit(".", 1)[0] not in col2im
            # a weight-norm g gradient is sum_k gW_k V_k / ||V|| per channel (asr/nn/convolution_2d.py:92): a signed sum of few hundred
            # terms, so whatever error gW has shows amplified by the cancellation
            tol = (LAYER_GX_COL2IM if upstream_of_col2im else LAYER_REL_L2) * (10 if pn.endswith(".g") else 1)
            assert err < tol, ("parameter gradient", pn, err)
    assert len(checked_grads) >= len(dev_grads) - 4, (len(checked_grads), len(dev_grads))      # all but the logit projection + its norm
    print("layer by layer %s/%d/wn=%s: worst forward %.1e (%s), input gradient %.1e (%s), parameter gradient %.1e (%s)"
          % (arch, nconv, wn, worst["y"][0], worst["y"][1], worst["gx"][0], worst["gx"][1], worst["gp"][0], worst["gp"][1]))
