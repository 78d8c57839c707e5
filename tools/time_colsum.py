import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rows, cols in [(32000, 640), (32000, 3000), (32000, 3072), (352000, 128), (1216000, 128)]:
    x = torch.randn(rows, cols, device=dev).to(torch.bfloat16)
    out = torch.zeros(cols, device=dev)
    ms = t(lambda: _ops.colsum_acc(x, out))
    print("colsum %d x %d: %.4f ms  %.2f TB/s" % (rows, cols, ms, rows * cols * 2 / ms / 1e9))
