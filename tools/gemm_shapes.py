"""Per-call shapes and device times of the NT / TN GEMMs in one train step (side stream off)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
from asr import _ops, functions as F
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
F._SIDE["enabled"] = False
for _ in range(3): step()
torch.cuda.synchronize()
log = []
o_nt, o_tn = _ops.gemm_nt, _ops.gemm_tn_acc
def nt(a, b, bias, od, out=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = o_nt(a, b, bias, od, out) if out is not None else o_nt(a, b, bias, od); e1.record()
    log.append(("nt", a.shape[0], b.shape[0], a.shape[1], str(od).replace("torch.", ""), e0, e1)); return r
def tn(a, b, c):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = o_tn(a, b, c); e1.record()
    log.append(("tn", a.shape[1], b.shape[1], a.shape[0], "f32acc", e0, e1)); return r
o_tng = _ops.gemm_tn_acc_group
def tng(products):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = o_tng(products); e1.record()
    fl = sum(2.0 * a.shape[1] * b.shape[1] * a.shape[0] for a, b, _ in products)
    log.append(("tn", -1, len(products), fl, " ".join("%dx%dx%d" % (a.shape[1], b.shape[1], a.shape[0]) for a, b, _ in products), e0, e1)); return r
_ops.gemm_nt, _ops.gemm_tn_acc = nt, tn
if _ops.TN_GROUP[0]: _ops.gemm_tn_acc_group = tng
step(); torch.cuda.synchronize()
tot = {"nt": 0.0, "tn": 0.0}
for kind, M, N, K, od, e0, e1 in log:
    ms = e0.elapsed_time(e1); tot[kind] += ms
    if M < 0:
        print("tn group of %d: %s  %.3f ms  %6.1f TF/s" % (N, od, ms, K / ms / 1e9)); continue
    print("%s M=%7d N=%5d K=%6d %-8s %.3f ms  %6.1f TF/s" % (kind, M, N, K, od, ms, 2.0 * M * N * K / ms / 1e9))
print(tot)
