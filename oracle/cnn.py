"""CPU oracle of the seven convolutional recipes of run/ctc/cnn/model.py:11-332 (torch-CPU fp32).  TEST INFRASTRUCTURE.

Written from the reference's ``build_model``: every recipe is a list of ``model.layer(...)`` blocks, a parameterised layer
takes the name ``layer_<index>`` (``layer_<index>_<inner>`` inside a Residual, asr/nn/nn.py:304-320), the index counting
EVERY entry of the blocks (links, lambdas, function layers).  ``program(arch, config)`` replays that bookkeeping and returns
the ops in order with the parameter names they read; ``forward`` executes them with stock torch operators in the
reference's logical layout (B, C, H, T).  conv / max-pool / maxout are Chainer's in the reference (absent: parity unpinned,
torch-CPU stand-in, see oracle/__init__.py); layer normalisation follows asr/nn/layernorm.py:33-48 (pinned by
tests/golden/norm.npz), GLU asr/nn/nn.py:277-281.
"""
import math

import torch
import torch.nn.functional as F

from . import nn as onn


def program(arch, cfg):
    """[(op, name or None, args)] for one recipe.  ops: conv (causal when pad_t > 0), ln, maxout, relu, pool, glu, res_begin,
    res_end.  `cfg`: vocab_size, ndim_audio_features, ndim_h, ndim_dense, num_conv_layers, kernel_size, num_mel_filters."""
    h, dense, V, cin, nconv = cfg.ndim_h, cfg.ndim_dense, cfg.vocab_size, cfg.ndim_audio_features, cfg.num_conv_layers
    kh, kw = cfg.kernel_size
    pad = kw - 1
    kernel_height = int(math.ceil((cfg.num_mel_filters - 2) / 3))          # run/ctc/cnn/model.py:38
    prog, index = [], [0]

    def block(entries, residual=False):
        """one model.layer(...) call; entries: (op, has_parameters, args).  A Residual is ONE entry of the block."""
        base = index[0]
        if residual:
            prog.append(("res_begin", None, None))
            for inner, (op, has_p, args) in enumerate(entries):
                prog.append((op, "layer_%d_%d" % (base, inner) if has_p else None, args))
            prog.append(("res_end", None, None))
            index[0] += 1
        else:
            for off, (op, has_p, args) in enumerate(entries):
                prog.append((op, "layer_%d" % (base + off) if has_p else None, args))
            index[0] += len(entries)

    conv = lambda ph: ("conv", True, (ph, pad))
    crop = ("crop", False, None)          # the lambda x: x[..., :-pad] entry (folded into the causal conv here)
    drop = ("dropout", False, None)
    maxout, relu, ln = ("maxout", False, None), ("relu", False, None), ("ln", True, None)
    point = ("conv", True, (0, 0))

    if arch in ("zhang", "zhang+fc_relu", "zhang+residual"):             # :40-204
        block([conv(0), crop, maxout, drop, ("pool", False, 3)])
        narrow, wide = min(nconv, 4), max(0, nconv - 4)
        for idx in range(narrow):
            entries = [conv(1), crop, maxout, drop]
            block(entries, residual=(arch == "zhang+residual" and idx != narrow - 1))      # :160-176
        for _ in range(narrow if wide > 0 else 0):                       # :64-72 / :178-187: the loop count is `narrow`
            block([conv(1), crop, maxout, drop], residual=(arch == "zhang+residual"))
        act = relu if arch == "zhang+fc_relu" else maxout
        block([point, act, drop])
        block([point, act, drop])
    elif arch == "zhang+layernorm":                                      # :206-236
        block([conv(0), crop, ln, maxout, drop, ("pool", False, 3)])
        for _ in range(nconv):
            block([conv(1), crop, ln, maxout, drop])
        block([point, ln, maxout, drop])
    elif arch == "glu":                                                  # :238-265
        block([conv(0), crop, maxout, drop, ("pool", False, 3)])
        for _ in range(nconv):
            block([("glu", True, (1, pad)), drop])
        block([("glu", True, (0, 0)), drop])
    elif arch == "relu+layernorm":                                       # :267-297
        block([conv(0), crop, ln, relu, drop, ("pool", False, 3)])
        for _ in range(nconv):
            block([conv(1), crop, ln, relu, drop])
        block([point, ln, relu, drop])
    elif arch == "relu+layernorm+residual":                              # :299-331
        block([conv(0), crop, ln, relu, drop, ("pool", False, 3)])
        for _ in range(nconv):
            block([ln, relu, drop, conv(1), crop], residual=True)
        block([point, ln, relu, drop])
    else:
        raise NotImplementedError(arch)
    block([point, ln])
    return prog


def _layer_norm(x, gamma, beta):
    """asr/nn/layernorm.py:33-48 + scale / bias on axis 1 (asr/nn/nn.py:260-265); no epsilon"""
    mean = x.mean(dim=(1, 2), keepdim=True)
    diff = x - mean
    std = torch.sqrt((diff * diff).mean(dim=(1, 2), keepdim=True))
    return diff / std * gamma.reshape(1, -1, 1, 1) + beta.reshape(1, -1, 1, 1)


def forward(arch, cfg, params, x, matched=False, fused_logit_bias=False, weights=None):
    """params: dict name -> torch tensor (requires_grad as the caller likes) with this package's state_dict names
    (``layer_0.W`` ...; a weight-normalised layer has ``.V``, ``.g``, ``.b``).  x (B, C, H, T) -> logits (B, V, 1, T).
    matched: bf16 roundings where the HIP path rounds (oracle/bf16.py): every operator output except the last projection and the
    last normalisation (float32 logits), every activation gradient an operator writes; fused_logit_bias: the last projection's
    bias gradient is the float32 column sum formed inside the fused LayerNorm + CTC sweep (V % 4 == 0).
    weights: {layer name: float32 W} -- the VALUE a weight-normalised layer's W takes in the forward pass (the device's own
    g V / (||V|| + 1e-9): its last float32 bit depends on the order of the norm's sum and decides bf16 roundings of W); gradients still
    flow through the formula."""
    from . import bf16 as Q
    prog = program(arch, cfg)
    h = Q.rnd(x) if matched else x
    return run(prog, 0, len(prog), params, h, matched, fused_logit_bias, weights)


def segments(prog):
    """[(lo, hi)] index ranges of `prog`, one per entry of the model's layer list: a Residual (res_begin .. res_end) is ONE entry"""
    out, i = [], 0
    while i < len(prog):
        if prog[i][0] == "res_begin":
            j = i
            while prog[j][0] != "res_end":
                j += 1
            out.append((i, j + 1))
            i = j + 1
        else:
            out.append((i, i + 1))
            i += 1
    return out


def run(prog, lo, hi, params, h, matched=False, fused_logit_bias=False, weights=None):
    """apply prog[lo:hi] to h (the ops of the whole recipe decide which projection / normalisation are the float32 logit layers)"""
    from . import bf16 as Q
    on = bool(matched)
    last_conv = max(i for i, (op, _, _) in enumerate(prog) if op in ("conv", "glu"))
    last_ln = max(i for i, (op, _, _) in enumerate(prog) if op == "ln")
    skip = None
    for i in range(lo, hi):
        op, name, args = prog[i]
        if op in ("conv", "glu"):
            ph, pt = args
            if name + ".V" in params:                                     # asr/nn/convolution_2d.py:21-25,62-64
                V = params[name + ".V"]
                W = params[name + ".g"] * V / (torch.sqrt((V * V).sum(dim=(1, 2, 3), keepdim=True)) + 1e-9)
                if weights is not None and name in weights:
                    W = W + (weights[name].reshape(W.shape) - W.detach())
            else:
                W = params[name + ".W"]
            b = params.get(name + ".b")
            f32_out = on and i == last_conv and last_ln > last_conv
            y = F.conv2d(Q.inp(h, on), Q.weight(W, on), None, stride=1, padding=(ph, pt))
            if pt > 0:
                y = y[..., :-pt]                                          # run/ctc/cnn/model.py:44, asr/nn/nn.py:276-277
            bias = 0.0 if b is None else b.reshape(1, -1, 1, 1)
            if f32_out:                                                   # float32 logits, gradient handed over in bf16
                y = (Q.inp(y, on) + bias) if fused_logit_bias else Q.inp(y + bias, on)
            else:
                y = Q.out(y + bias, on)
            if op == "glu":
                a, g = torch.chunk(Q.inp(y, on), 2, dim=1)                # asr/nn/nn.py:279-280
                y = Q.out(a * torch.sigmoid(g), on)
            h = y
        elif op == "ln":
            y = _layer_norm(Q.inp(h, on), params[name + ".gamma"], params[name + ".beta"])
            h = y if (on and i == last_ln) else Q.out(y, on)
        elif op == "maxout":
            h = Q.out(onn.maxout2(h), on)
        elif op == "relu":
            h = Q.out(torch.relu(Q.inp(h, on)), on)
        elif op == "pool":
            h = Q.out(onn.maxpool_h(h, args), on)
        elif op == "res_begin":
            skip = h
        elif op == "res_end":
            h = Q.out(h + skip, on)                                       # asr/nn/nn.py:322-328
    return h


def logits_tbv(out):
    """(B, V, 1, T) -> (T, B, V): asr/model/cnn.py:41-44"""
    return out[:, :, 0, :].permute(2, 0, 1)
