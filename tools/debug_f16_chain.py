"""Free-running (not teacher-forced) comparison of a recipe with the rounding-matched oracle: per layer, the forward output and the
gradient arriving at that output, device against oracle -- where does the end-to-end difference enter?  Either build (ASR_ACT)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from asr.model import cnn
from asr.model.architectures import build_model
from asr.loss import connectionist_temporal_classification
from asr import functions as F
from oracle import model as omodel, cnn as ocnn
import test_model_gpu as tm

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
ys, rec = tm._trace_layers(model, xd)
loss = connectionist_temporal_classification(ys, labels.to(dev), 0, x_len.to(dev), l_len.to(dev))
loss.backward()
F.join_side_stream(); torch.cuda.synchronize()
prog = ocnn.program(arch, cfg)
segs = ocnn.segments(prog)
params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.named_parameters()}
from oracle import bf16 as Q
h = Q.rnd(x)
outs = []
for lo, hi in segs:
    h = ocnn.run(prog, lo, hi, params, h, matched=True, fused_logit_bias=False)
    h.retain_grad()
    outs.append(h)
omodel.ctc_mean_loss(ocnn.logits_tbv(h), labels, x_len, l_len).backward()


def rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def to_nchw(t, like):       # the device's tensors are views of its physical layout with the logical (B, C, H, T) shape
    return t.float().cpu().reshape(like.shape)


print("act", model.layers and next(iter(rec))["xout"].dtype)
for k, e in enumerate(rec):
    o = outs[e["span"][1] - 1]
    ops = [p[0] + ":" + str(p[1]) for p in prog[segs[e["span"][0]][0]:segs[e["span"][1] - 1][1]]]
    fy = rel(to_nchw(e["xout"], o), o.detach())
    gy = rel(to_nchw(e["gout"], o), o.grad) if e["gout"] is not None and o.grad is not None else float("nan")
    ulps = ""
    if e["gout"] is not None and o.grad is not None:
        d = (to_nchw(e["gout"], o) - o.grad).abs()
        ulps = "differing elements %.1f%%, max|g| %.3g" % (100.0 * float((d > 0).float().mean()), float(o.grad.abs().max()))
    print("layer %2d %-60s forward %.2e   gradient at output %.2e   %s" % (k, " ".join(ops)[:60], fy, gy, ulps))
for n, p in model.named_parameters():
    print("   %-14s %.2e" % (n, rel(p.grad.cpu(), params[n].grad)))
