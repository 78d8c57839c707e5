"""Quick timing of the full train step at the BASELINE shape (used while developing; bench.py is the contract)."""
import sys, os, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch
from asr.model import ds2
from asr.loss import connectionist_temporal_classification
from asr.optimizers import Adam, GradientClipping, WeightDecay
from asr.data.synthetic import synthetic_batch

def main(B=32, T=1000, V=3000, steps=5):
    dev = torch.device("cuda:0")
    cfg = ds2.configure(); cfg.vocab_size = V
    torch.manual_seed(0)
    model = ds2.Model(cfg).to_gpu()
    x, labels, x_len, l_len = synthetic_batch(B, T, V)
    x, labels, x_len, l_len = x.to(dev), labels.to(dev), x_len.to(dev), l_len.to(dev)
    opt = Adam(1e-3, 0.9); opt.setup(model); opt.add_hook(GradientClipping(1.0)); opt.add_hook(WeightDecay(1e-5))
    def step():
        loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
        opt.update(lossfun=lambda: loss)
        return loss
    for i in range(2):
        l = step(); torch.cuda.synchronize(); print("warm", i, l.item(), flush=True)
    t0 = time.time()
    for i in range(steps):
        l = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    print(json.dumps(dict(ms_per_step=dt * 1e3, utt_per_s=B / dt, loss=l.item(), params=sum(p.numel() for p in model.parameters()),
                          mem_GB=torch.cuda.max_memory_allocated() / 2**30)))

if __name__ == "__main__":
    main()
