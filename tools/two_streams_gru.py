"""Two half-batch GRU launches on two streams: where do the recurrences land (XCC ids) and do the launches overlap?"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
from asr.pipeline import HalfBatches
T, B, H, ndir = 1000, 16, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
whh16 = whh.to(torch.bfloat16).contiguous()
whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
bhh = torch.zeros(ndir * 3 * H, device=dev)
dy = (torch.randn(T * B, H, generator=g) * 0.1).to(dev).to(torch.bfloat16)
dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
pipe = HalfBatches(dev)
def run(which, both):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    outs, syncs = [], []
    for i in ((0, 1) if both else (0,)):
        with pipe.half(i):
            for _ in range(3):
                if which == "fwd":
                    o = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
                else:
                    o = _ops.gru_bwd(dy, saved[2], saved[0], whhT16, T, B, H, ndir, dbi, dbh)
            syncs.append(_ops.LAST_SYNC[0])
    pipe.join()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 3, syncs
y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
saved = (hseq, hseq16, gates)
for which in ("fwd", "bwd"):
    for both in (False, True):
        run(which, both)
        ms, syncs = run(which, both)
        print(which, "both streams" if both else "one stream ", "%.3f ms per launch round" % ms)
        for s in syncs:
            w = s[960:1008].cpu().tolist()
            print("   xcc", w[0:8], "mismatch", w[16:24], "arrivals", w[32:40])
_ops.gru_check_sync()
