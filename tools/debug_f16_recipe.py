"""Half build: where does a recipe's backward pass leave the half range?  ASR_ACT=f16 python tools/debug_f16_recipe.py [arch] [nconv]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch
from asr.model import cnn
from asr.model.architectures import build_model
from asr.loss import connectionist_temporal_classification
from asr.functions import join_side_stream
from oracle import model as omodel

arch = sys.argv[1] if len(sys.argv) > 1 else "glu"
nconv = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = cnn.configure()
cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers, cfg.architecture = 19, 3, 16, 24, nconv, arch
model = build_model(cfg).to_gpu()
x, labels, x_len, l_len = omodel.synthetic_batch(3, 40, 19, Lmin=2, Lmax=6, seed=5)
xd, ld, xl, ll = x.to(dev), labels.to(dev), x_len.to(dev), l_len.to(dev)
with torch.no_grad():
    model(xd)
for seed in (1.0, 16.0, 256.0, 4096.0):
    for p in model.parameters():
        p.grad = None
    seen = []
    loss = connectionist_temporal_classification(model(xd), ld, 0, xl, ll)
    nodes, stack, order = set(), [loss.grad_fn], []
    while stack:
        n = stack.pop()
        if n is None or n in nodes:
            continue
        nodes.add(n)
        order.append(n)
        stack.extend(f for f, _ in n.next_functions)

    def mx(ts):
        out = []
        for t in ts:
            if torch.is_tensor(t) and t.numel():
                f = t.float()
                out.append("%s%s %.3g%s" % (str(t.dtype).replace("torch.", ""), list(t.shape), float(f[torch.isfinite(f)].abs().max()) if torch.isfinite(f).any() else float("nan"),
                                            "" if torch.isfinite(f).all() else " NON-FINITE x%d" % int((~torch.isfinite(f)).sum())))
        return "; ".join(out)
    for n in order:
        n.register_hook(lambda gi, go, n=n: seen.append((type(n).__name__, "in: " + mx(go), "out: " + mx(gi))))
    loss.backward(gradient=torch.full_like(loss, seed))
    join_side_stream(); torch.cuda.synchronize()
    bad = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print("seed %g loss %.3f: non-finite parameter gradients: %s" % (seed, loss.item(), bad))
    for s in seen:
        print("    %-26s %s | %s" % s)
