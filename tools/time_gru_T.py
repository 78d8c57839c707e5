"""Launch overhead of the recurrence kernels: time against T (intercept = start-up + drain): python tools/time_gru_T.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
B, H, ndir = 32, 512, 2
res = []
for T in (4, 50, 100, 200, 400, 1000):
    g = torch.Generator().manual_seed(0)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(torch.bfloat16)
    whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
    whh16 = whh.to(torch.bfloat16).contiguous()
    whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
    bhh = torch.zeros(ndir * 3 * H, device=dev)
    dy = torch.randn(T * B, H, generator=g).to(dev).to(torch.bfloat16)
    dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
    y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
    f = t(lambda: _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)) * 1e3
    b = t(lambda: _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)) * 1e3
    res.append((T, f, b))
    print("T=%d: fwd %.1f us, bwd %.1f us" % (T, f, b))
(T0, f0, b0), (T1, f1, b1) = res[1], res[-1]
sf, sb = (f1 - f0) / (T1 - T0), (b1 - b0) / (T1 - T0)
print("slope fwd %.3f bwd %.3f us/step; intercept fwd %.1f bwd %.1f us per launch (incl. sync clear / fill / merge launches)" % (sf, sb, f0 - sf * T0, b0 - sb * T0))
