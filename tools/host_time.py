"""Host-side enqueue time of a train step against its device time: how far ahead of the GPU the Python side runs."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
from asr.loss import connectionist_temporal_classification
from asr.model import ds2
from asr.optimizers import Adam, GradientClipping, WeightDecay
from asr.data.synthetic import synthetic_batch
dev = torch.device("cuda:0")
cfg = ds2.configure(); cfg.vocab_size = 3000
torch.manual_seed(0)
model = ds2.Model(cfg).to_gpu(0)
x, labels, x_len, l_len = [t.to(dev) for t in synthetic_batch(32, 1000, 3000, seed=0)]
opt = Adam(alpha=1e-3, beta1=0.9); opt.setup(model); opt.add_hook(GradientClipping(1.0)); opt.add_hook(WeightDecay(1e-5))
def step():
    loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
    opt.update(lossfun=lambda: loss)
for _ in range(5): step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue %.2f ms per step; device %.2f ms per step (the host ran %.1f ms ahead at the end)" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3, (t2 - t1) * 1e3))
# one step alone, from an idle device: host time until everything is queued
torch.cuda.synchronize(); t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("single step from idle: queued after %.2f ms, done after %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
