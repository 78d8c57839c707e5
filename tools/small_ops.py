"""Which torch-level operators (not our C-ABI kernels) a train step still launches, with their device time: small launches on the
launch stream each cost 4-9 us of a step that is otherwise one dependent chain."""
import sys, os, torch
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
for _ in range(3): step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key.startswith("aten::") and e.device_time_total > 0]
rows.sort(key=lambda e: -e.device_time_total)
for e in rows[:40]:
    where = [s for s in e.stack if "asr/" in s or "bench" in s or "tools/" in s][:2]
    print("%-28s x%-3d %7.1f us  %s" % (e.key, e.count, e.device_time_total, " <- ".join(w.strip()[-70:] for w in where)))
