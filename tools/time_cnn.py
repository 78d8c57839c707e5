"""Train-step time of a CNN recipe of run/ctc/cnn/model.py (BASELINE configs[4]: zhang+residual, ndim_h=128,
ndim_dense=320, V=119) on one GPU -- a parity-test configuration, timed here for reference only."""
import sys, os, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
from asr.model import cnn
from asr.model.architectures import build_model
from asr.loss import connectionist_temporal_classification
from asr.optimizers import get_optimizer, GradientClipping, WeightDecay
from asr.data.synthetic import synthetic_batch
arch = sys.argv[1] if len(sys.argv) > 1 else "zhang+residual"
B, T, V = 32, 1000, 119
dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = cnn.configure()
cfg.vocab_size, cfg.ndim_audio_features, cfg.ndim_h, cfg.ndim_dense, cfg.num_conv_layers, cfg.architecture = V, 3, 128, 320, 4, arch
model = build_model(cfg).to_gpu()
x, labels, x_len, l_len = [t.to(dev) for t in synthetic_batch(B, T, V, seed=0)]
opt = get_optimizer("adam", 1e-3, 0.9)
ys = model(x)
opt.setup(model); opt.add_hook(GradientClipping(1.0)); opt.add_hook(WeightDecay(1e-5))
def step():
    loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
    opt.update(lossfun=lambda: loss)
    return loss
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): loss = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
nparam = sum(p.numel() for p in model.parameters())
print("%s: %.2f ms/step, %.0f utt/s, %.1f M parameters, loss %.3f, peak memory %.1f GB" % (arch, dt * 1e3, B / dt, nparam / 1e6, loss.item(), torch.cuda.max_memory_allocated() / 2**30))
