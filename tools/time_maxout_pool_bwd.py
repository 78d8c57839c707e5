"""maxout(2) + pooling backward on the first conv block of the BASELINE model, with and without the bias-gradient sums."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (T, B, H, C, k) in [(1000, 32, 38, 64, 3), (1000, 32, 11, 64, 2)]:
    x = torch.randn(T, B, H, 2 * C, device=dev).to(torch.bfloat16)
    y = _ops.maxout2_pool_fwd(x, k)
    gy = torch.randn(y.shape, device=dev).to(torch.bfloat16)
    db = torch.zeros(2 * C, device=dev)
    print("H=%d plain %.1f us, with db %.1f us, colsum alone %.1f us" % (H, 1e3 * t(lambda: _ops.maxout2_pool_bwd(x, gy, k)),
          1e3 * t(lambda: _ops.maxout2_pool_bwd(x, gy, k, db)), 1e3 * t(lambda: _ops.colsum_acc(x.reshape(-1, 2 * C), db))))
