"""Forward / backward recurrence per time step over a few shapes (default mode): python tools/time_gru_shapes.py"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
def t(fn, n=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(1000, 32, 512, 2), (1000, 32, 256, 2), (1000, 24, 512, 2), (1000, 32, 512, 1), (1000, 32, 384, 2), (1000, 32, 128, 2)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for (T, B, H, ndir) in shapes:
    g = torch.Generator().manual_seed(0)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
    gi = gi.to(_ops.gru_gi_dtype(T, B, H, ndir))
    whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
    whh16 = whh.to(torch.bfloat16).contiguous()
    whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
    bhh = torch.zeros(ndir * 3 * H, device=dev)
    dy = torch.randn(T * B, H, generator=g).to(dev).to(torch.bfloat16)
    dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
    y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
    f = t(lambda: _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)) / T * 1e3
    b = t(lambda: _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)) / T * 1e3
    _ops.gru_check_sync()
    print("[delay=%s] T=%d B=%d H=%d ndir=%d: fwd %.3f us/step, bwd %.3f us/step" % (os.environ.get("ASR_DEBUG", "auto"), T, B, H, ndir, f, b))
