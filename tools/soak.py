"""Soak: N free-running train steps of the BASELINE configs[1] model on a fixed synthetic ragged batch -- every step applied (no
recurrence gave up, no non-finite gradient), loss falling, step time flat.  python tools/soak.py [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch
from asr import _ops
from asr.loss import connectionist_temporal_classification
from asr.model import ds2
from asr.optimizers import Adam, GradientClipping, WeightDecay
from asr.data.synthetic import synthetic_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = ds2.configure(); cfg.vocab_size = 3000
model = ds2.Model(cfg).to_gpu(0)
x, labels, x_len, l_len = (t.to(dev) for t in synthetic_batch(32, 1000, 3000, seed=0, ragged=True))
opt = Adam(alpha=3e-4, beta1=0.9); opt.setup(model); opt.add_hook(GradientClipping(1.0)); opt.add_hook(WeightDecay(1e-5))
losses, marks = [], []
t0 = time.perf_counter()
for i in range(N):
    loss = connectionist_temporal_classification(model(x, x_length=x_len), labels, 0, x_len, l_len)
    opt.update(lossfun=lambda: loss)
    if i % 100 == 0 or i == N - 1:
        losses.append((i, loss))
        ev = torch.cuda.Event(enable_timing=True); ev.record(); marks.append((i, ev))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("steps %d in %.2f s (%.2f ms per step), applied %d" % (N, dt, dt / N * 1e3, opt.applied_steps()))
print("loss:", ", ".join("%d: %.1f" % (i, float(l.detach())) for i, l in losses))
print("ms per step between marks:", ", ".join("%.2f" % (marks[k][1].elapsed_time(marks[k + 1][1]) / (marks[k + 1][0] - marks[k][0])) for k in range(len(marks) - 1)))
_ops.gru_check_all()
assert opt.applied_steps() == N and all(bool(torch.isfinite(l.detach())) for _, l in losses) and float(losses[-1][1].detach()) < float(losses[0][1].detach())
print("ok")
