"""Debug aid: per-layer forward activations and per-parameter gradients of a CNN recipe, device vs the rounding-matched oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch
from asr.model import cnn
from asr.model.architectures import build_model
from asr.loss import connectionist_temporal_classification
from asr.nn import nn as nnmod
from asr import functions as F
from oracle import cnn as ocnn, model as omodel, bf16 as Q, nn as onn
import torch.nn.functional as TF

dev = torch.device("cuda:0")

def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))

def run(arch, nconv, wn, V=19, B=3, T=36):
    torch.manual_seed(3)
    cfg = cnn.configure()
    cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers = V, 3, 16, 24, nconv
    cfg.architecture, cfg.weightnorm = arch, wn
    model = build_model(cfg).to_gpu()
    x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=2, Lmax=6, seed=7, ragged=True)
    xd = x.to(dev)
    with torch.no_grad():
        model(xd)
    trace_dev = []
    orig = nnmod._apply_layers
    def traced(layers, xx):
        i = 0
        while i < len(layers):
            layer = layers[i]
            j = nnmod._fusable_pool(layers, i) if xx.dim() == 4 else -1
            if j > 0:
                ks = layers[j].ksize
                xx = F.maxout_max_pooling(xx, ks[0] if isinstance(ks, (tuple, list)) else ks, sole_consumer=i > 0)
                trace_dev.append(("maxout+pool", xx.detach().float().cpu()))
                i = j + 1
                continue
            y = layer(xx)
            if isinstance(layer, nnmod.Residual):
                y = F.add(y, xx)
            xx = y
            trace_dev.append((type(layer).__name__, xx.detach().float().cpu()))
            i += 1
        return xx
    nnmod._apply_layers = traced
    try:
        ys = model(xd)
    finally:
        nnmod._apply_layers = orig
    loss = connectionist_temporal_classification(ys, labels.to(dev), 0, x_len.to(dev), l_len.to(dev))
    loss.backward()
    F.join_side_stream(); torch.cuda.synchronize()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
    # oracle with trace
    trace_or = []
    on = True
    prog = ocnn.program(arch, cfg)
    last_conv = max(i for i, (op, _, _) in enumerate(prog) if op in ("conv", "glu"))
    last_ln = max(i for i, (op, _, _) in enumerate(prog) if op == "ln")
    h, skip = Q.rnd(x), None
    for i, (op, name, args) in enumerate(prog):
        if op in ("conv", "glu"):
            ph, pt = args
            if name + ".V" in params:
                Vv = params[name + ".V"]
                W = params[name + ".g"] * Vv / (torch.sqrt((Vv * Vv).sum(dim=(1, 2, 3), keepdim=True)) + 1e-9)
            else:
                W = params[name + ".W"]
            b = params.get(name + ".b")
            f32_out = i == last_conv and last_ln > last_conv
            y = TF.conv2d(Q.inp(h), Q.weight(W), None, stride=1, padding=(ph, pt))
            if pt > 0:
                y = y[..., :-pt]
            bias = 0.0 if b is None else b.reshape(1, -1, 1, 1)
            y = Q.inp(y + bias) if f32_out else Q.out(y + bias)
            h = y
        elif op == "ln":
            y = ocnn._layer_norm(Q.inp(h), params[name + ".gamma"], params[name + ".beta"])
            h = y if i == last_ln else Q.out(y)
        elif op == "maxout":
            h = Q.out(onn.maxout2(h))
        elif op == "relu":
            h = Q.out(torch.relu(Q.inp(h)))
        elif op == "pool":
            h = Q.out(onn.maxpool_h(h, args))
        elif op == "res_begin":
            skip = h
        elif op == "res_end":
            h = Q.out(h + skip)
        trace_or.append((op + ":" + str(name), h.detach()))
    loss_ref = omodel.ctc_mean_loss(ocnn.logits_tbv(h), labels, x_len, l_len)
    loss_ref.backward()
    print("=====", arch, nconv, wn, "loss", loss.item(), loss_ref.item())
    # align: walk both traces; a device "maxout+pool" consumes oracle entries up to its pool, a device Residual up to its res_end
    k = 0
    for name, t in trace_dev:
        if name == "maxout+pool":
            while not trace_or[k][0].startswith("pool"):
                k += 1
        elif name == "Residual":
            while not trace_or[k][0].startswith("res_end"):
                k += 1
        else:
            while trace_or[k][0].split(":")[0] in ("res_begin",):
                k += 1
        o = trace_or[k]
        k += 1
        if tuple(o[1].shape) != tuple(t.shape):
            print("  %-18s %-22s SHAPE MISMATCH %s %s" % (name, o[0], tuple(t.shape), tuple(o[1].shape)))
            continue
        print("  %-18s %-22s %s rel %.2e  max|dev| %.3g" % (name, o[0], tuple(t.shape), rel(t, o[1]), float(t.abs().max())))
    for name, p in model.named_parameters():
        print("   grad %-16s %-18s rel %.2e  |g| %.3e" % (name, tuple(p.shape), rel(p.grad.cpu(), params[name].grad), float(params[name].grad.norm())))

for arch, nconv, wn in (("zhang+residual", 4, True), ("zhang", 3, True), ("zhang+residual", 6, False)):
    run(arch, nconv, wn)
