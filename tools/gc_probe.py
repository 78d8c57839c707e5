"""Where do the occasional 30 - 55 ms steps come from?  300 free-running train steps of the bench model with a callback on every garbage
collection (generation, duration) and the host's enqueue time per step; prints every collection > 2 ms and every host step > 8 ms."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch
from asr.loss import connectionist_temporal_classification
from asr.model import ds2
from asr.optimizers import Adam, GradientClipping, WeightDecay
from asr.data.synthetic import synthetic_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = ds2.configure(); cfg.vocab_size = 3000
model = ds2.Model(cfg).to_gpu(0)
x, labels, x_len, l_len = (t.to(dev) for t in synthetic_batch(32, 1000, 3000, seed=0))
opt = Adam(alpha=1e-3, beta1=0.9); opt.setup(model); opt.add_hook(GradientClipping(1.0)); opt.add_hook(WeightDecay(1e-5))
events, t_gc = [], [0.0]
def cb(phase, info):
    if phase == "start": t_gc[0] = time.perf_counter()
    else: events.append((step_no[0], info["generation"], (time.perf_counter() - t_gc[0]) * 1e3, info.get("collected", 0)))
gc.callbacks.append(cb)
step_no = [0]
freeze = len(sys.argv) > 1 and sys.argv[1] == "freeze"
host = []
for i in range(300):
    if i == 5:
        torch.cuda.synchronize()
        if freeze:
            gc.collect(); gc.freeze()
    step_no[0] = i
    t0 = time.perf_counter()
    loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
    opt.update(lossfun=lambda: loss)
    host.append((time.perf_counter() - t0) * 1e3)
torch.cuda.synchronize()
print("freeze" if freeze else "default", "gc counts", gc.get_count(), "thresholds", gc.get_threshold(), "objects tracked", len(gc.get_objects()))
print("collections > 2 ms:", [(s, g, round(ms, 1), c) for s, g, ms, c in events if ms > 2.0])
print("by generation:", {g: (sum(1 for e in events if e[1] == g), round(sum(e[2] for e in events if e[1] == g), 1)) for g in (0, 1, 2)})
hs = sorted(host[5:])
print("host ms per step: median %.2f p99 %.2f max %.2f; steps > 8 ms: %s" % (hs[len(hs) // 2], hs[int(len(hs) * 0.99)], hs[-1], [(i, round(h, 1)) for i, h in enumerate(host) if h > 8.0 and i >= 5]))
