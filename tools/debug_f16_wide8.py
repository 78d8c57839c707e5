"""Half build, configs[4] wide recipe at full size: loss, loss scale and the largest activation per step."""
import os, sys, copy, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from asr.nn import nn as nnmod
from asr.model.architectures import build_model
from asr.loss import connectionist_temporal_classification
from asr.optimizers import Adam, GradientClipping, WeightDecay
from asr.data.synthetic import synthetic_batch

nconv = int(sys.argv[1]) if len(sys.argv) > 1 else 8
scale0 = float(sys.argv[2]) if len(sys.argv) > 2 else None
dev = torch.device("cuda:0")
a = argparse.Namespace(num_conv_layers=nconv, frames=1000)
for k, v in vars(bench.parse.__globals__.get("DEFAULTS", {})).items() if False else []:
    pass
sys.argv = [sys.argv[0]]
args = bench.parse()
args.num_conv_layers = nconv
B, T, V = 32, 1000, 119
torch.manual_seed(0)
cfg = bench.cnn_config(args, V)
model = build_model(cfg).to_gpu(0)
x, labels, x_len, l_len = (t.to(dev) for t in synthetic_batch(B, T, V, seed=0))
with torch.no_grad():
    model(x)
opt = Adam(alpha=1e-3, beta1=0.9)
opt.setup(model)
opt.add_hook(GradientClipping(1.0))
opt.add_hook(WeightDecay(1e-5))
if scale0 is None:
    opt.loss_scaling()
else:
    opt.loss_scaling(scale=scale0)
orig = nnmod._apply_layers
peak = []


def traced(layers, x, *a, **k):
    for i in range(len(layers)):
        x = orig(layers[i:i + 1], x, *a, **k)
        if torch.is_tensor(x):
            f = x.detach().float()
            peak.append((i, type(layers[i]).__name__, float(f[torch.isfinite(f)].abs().max()) if torch.isfinite(f).any() else float("nan"), int((~torch.isfinite(f)).sum())))
    return x


for step in range(20):
    peak.clear()
    if step in (0, 3, 19):
        nnmod._apply_layers = traced
    try:
        loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
    finally:
        nnmod._apply_layers = orig
    opt.update(lossfun=lambda: loss)
    s, o = opt.loss_scale()
    print("step %2d loss %.4f scale %g overflows %d applied %d |g|^2 %.3g" % (step, loss.item(), s, o, opt.applied_steps(), float(opt._flat["ctl"][4])), flush=True)
    if peak:
        print("    largest |activation| per top-level layer:", ", ".join("%d:%s %.3g%s" % (i, n[:8], m, "" if not bad else " (%d non-finite)" % bad) for i, n, m, bad in peak))
