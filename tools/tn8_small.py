"""the dense layers' weight gradients (few tiles, K = 32000: split-K float atomics dominate) on the eight-wave TN kernel: python tools/tn8_small.py
(ASR_DEBUG tn8_items=N / tn8_stag=n are read once per process: run it once per setting)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
out = []
for K, M, N in [(32000, 640, 512), (32000, 640, 320), (32000, 3000, 320), (32000, 512, 640), (352000, 128, 960)]:
    a = torch.randn(K, M).to(dev).to(BF16)
    b = torch.randn(K, N).to(dev).to(BF16)
    c = torch.zeros(M, N, device=dev)
    for _ in range(5):
        _ops.gemm_tn_acc_group_8ph([(a, b, c)])
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _ops.gemm_tn_acc_group_8ph([(a, b, c)])
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    out.append("%dx%d: %.1f" % (M, N, best))
print("ASR_DEBUG=%-28s %s" % (os.environ.get("ASR_DEBUG", ""), "   ".join(out)))
