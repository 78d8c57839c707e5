"""the model's short- and long-K NT products through the routed entry asr_gemm_nt: python tools/time_nt_routed.py (ASR_DEBUG is read once per process)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chainer-speech-recognition_amd"))
import torch
from asr import _ops
dev = torch.device("cuda", 0)
out = []
for M, N, K in [(32000, 640, 320), (32000, 640, 512), (32000, 320, 640), (32000, 512, 640), (32000, 3072, 512), (32000, 384, 3072)]:
    a = torch.randn(M, K).to(dev).to(torch.bfloat16)
    b = torch.randn(N, K).to(dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    for _ in range(5):
        _ops.gemm_nt(a, b, bias, torch.bfloat16)
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _ops.gemm_nt(a, b, bias, torch.bfloat16)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    out.append("%dx%dx%d: %.1f" % (M, N, K, best))
print("ASR_DEBUG=%-12s %s" % (os.environ.get("ASR_DEBUG", ""), "  ".join(out)))
