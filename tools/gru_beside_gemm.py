"""Does a weight-gradient GEMM on the OTHER CUs slow a half-chip recurrence?  B=16 backward recurrence (8 recurrences x 16 workgroups =
128 CUs, 16 per XCD) with and without TN GEMMs running on a second stream."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 16, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(torch.bfloat16)
whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
whh16 = whh.to(torch.bfloat16).contiguous()
whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
bhh = torch.zeros(ndir * 3 * H, device=dev)
dy = (torch.randn(T * B, H, generator=g) * 0.1).to(dev).to(torch.bfloat16)
dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
y, hseq, hseq16, gates = _ops.gru_fwd(gi.float() if _ops.gru_gi_dtype(T, B, H, ndir) != torch.bfloat16 else gi, whh16, bhh, T, B, H, ndir)
K, M, N = 32000, 3072, 512
a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
c = torch.zeros(M, N, device=dev)
side = torch.cuda.Stream()
def rec():
    return _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)
def timed(fn, stream=None):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream); fn(); e1.record(stream)
    return e0, e1
rec(); _ops.gemm_tn_acc(a, b, c); torch.cuda.synchronize()
e0, e1 = timed(rec); torch.cuda.synchronize(); t_rec = e0.elapsed_time(e1)
e0, e1 = timed(lambda: [_ops.gemm_tn_acc(a, b, c) for _ in range(10)]); torch.cuda.synchronize(); t_gemm = e0.elapsed_time(e1) / 10
print("recurrence alone %.3f ms (%.3f us/step), TN GEMM alone %.3f ms" % (t_rec, t_rec / T * 1e3, t_gemm))
for order in ("gemms first", "recurrence first"):
    torch.cuda.synchronize()
    n = 12
    def gemms():
        with torch.cuda.stream(side):
            return timed(lambda: [_ops.gemm_tn_acc(a, b, c) for _ in range(n)], side)
    if order == "gemms first":
        g0, g1 = gemms(); r0, r1 = timed(rec)
    else:
        r0, r1 = timed(rec); g0, g1 = gemms()
    torch.cuda.synchronize()
    print("%s: recurrence %.3f ms (%.3f us/step), %d GEMMs %.3f ms (%.3f each; alone %.3f); end-to-end %.3f ms" % (
        order, r0.elapsed_time(r1), r0.elapsed_time(r1) / T * 1e3, n, g0.elapsed_time(g1), g0.elapsed_time(g1) / n, t_gemm,
        max(g0.elapsed_time(g1), r0.elapsed_time(r1))))
_ops.gru_check_sync()
