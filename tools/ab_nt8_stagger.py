"""short-K NT products on the eight-wave kernel under ASR_DEBUG nt8_stagger=<n> (start delay of the first round's workgroups)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
out = []
for M, N, K, od in [(32000, 3072, 512, BF16), (32000, 3072, 384, BF16), (32000, 3000, 320, torch.float32), (32000, 512, 640, BF16)]:
    a = torch.randn(M, K).to(dev).to(BF16)
    b = torch.randn(N, K).to(dev).to(BF16)
    c = torch.empty(M, N, dtype=od, device=dev)
    for _ in range(5):
        _ops.gemm_nt_8ph(a, b, None, od, c)
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _ops.gemm_nt_8ph(a, b, None, od, c)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    out.append("%dx%dx%d %.1f" % (M, N, K, best))
print("ASR_DEBUG=%-16s" % os.environ.get("ASR_DEBUG", ""), " | ".join(out))
