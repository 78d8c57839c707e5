"""Half build: parameter-gradient error of a recipe against the rounding-matched oracle as a function of the backward seed.
ASR_ACT=f16 python tools/debug_f16_seed.py [arch] [nconv]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch
from asr.model import cnn
from asr.model.architectures import build_model
from asr.loss import connectionist_temporal_classification
from asr.functions import join_side_stream
from oracle import model as omodel, cnn as ocnn

arch = sys.argv[1] if len(sys.argv) > 1 else "zhang+residual"
nconv = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda:0")
torch.manual_seed(3)
V, B, T = 19, 3, 36
cfg = cnn.configure()
cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers, cfg.architecture = V, 3, 16, 24, nconv, arch
model = build_model(cfg).to_gpu()
x, labels, x_len, l_len = omodel.synthetic_batch(B, T, V, Lmin=2, Lmax=6, seed=7, ragged=True)
xd = x.to(dev)
with torch.no_grad():
    model(xd)


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


for seed in (2.0 ** -14, 2.0 ** -8, 1.0, 1024.0):
    for p in model.parameters():
        p.grad = None
    loss = connectionist_temporal_classification(model(xd), labels.to(dev), 0, x_len.to(dev), l_len.to(dev))
    loss.backward(gradient=torch.full_like(loss, seed))
    join_side_stream(); torch.cuda.synchronize()
    line = []
    for matched in (True, False):
        params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
        out = ocnn.forward(arch, cfg, params, x, matched=matched, fused_logit_bias=False)
        (omodel.ctc_mean_loss(ocnn.logits_tbv(out), labels, x_len, l_len) * seed).backward()
        errs = {n: rel(p.grad.cpu(), params[n].grad) for n, p in model.named_parameters()}
        w = max(errs, key=errs.get)
        if matched and seed == 1.0:
            print("   matched, every parameter:", ", ".join("%s %.1e" % kv for kv in errs.items()))
        line.append("%s oracle worst %.2e (%s), layer_0.W %.2e" % ("matched" if matched else "float32", errs[w], w, errs.get("layer_0.W", 0.0)))
    print("seed %7g: %s" % (seed, "; ".join(line)), flush=True)
