"""one library's asr_gemm_nt_8ph on the model's NT shapes (to be run alternately with ASR_HIP_LIB pointing at two builds, on one box)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
out = []
for M, N, K in [(32000, 512, 3072), (32000, 3072, 512), (4096, 4096, 4096), (32000, 384, 3072)]:
    a = torch.randn(M, K).to(dev).to(BF16)
    b = torch.randn(N, K).to(dev).to(BF16)
    c = torch.empty(M, N, dtype=BF16, device=dev)
    for _ in range(5):
        _ops.gemm_nt_8ph(a, b, None, BF16, c)
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _ops.gemm_nt_8ph(a, b, None, BF16, c)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    out.append("%dx%dx%d %.1f us" % (M, N, K, best))
print(os.environ.get("ASR_HIP_LIB", "in-tree")[-28:], " | ".join(out))
