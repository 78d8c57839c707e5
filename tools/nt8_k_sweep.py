"""eight-wave NT kernel, 32000 x 3072 x K for a sweep of K: the fixed cost of a tile (fill latency, epilogue, workgroup turnover) against its K loop"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
M, N = 32000, 3072
for K in (64, 128, 256, 512, 1024, 2048):
    a = torch.randn(M, K).to(dev).to(BF16)
    b = torch.randn(N, K).to(dev).to(BF16)
    c = torch.empty(M, N, dtype=BF16, device=dev)
    for _ in range(5):
        _ops.gemm_nt_8ph(a, b, None, BF16, c)
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _ops.gemm_nt_8ph(a, b, None, BF16, c)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print("K=%5d: %.1f us  (%.1f us per round of 256 tiles, %.0f TFLOP/s)" % (K, best, best / (1500 / 256.0), 2.0 * M * N * K / best / 1e6))
