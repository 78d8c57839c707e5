import sys, os, torch
sys.path.insert(0, "/root/repo/chainer-speech-recognition_amd"); sys.path.insert(0, "/root/repo")
from asr import _lib, _ops, functions as F
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
