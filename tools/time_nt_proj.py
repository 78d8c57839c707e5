"""Projection-shaped NT GEMMs of the step (ASR_DEBUG nt_persist=0 / 1 selects the kernel): python tools/time_nt_proj.py"""
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
tag = os.environ.get("ASR_DEBUG", "default")
for (M, N, K, od) in [(32000, 3072, 512, torch.bfloat16), (32000, 3072, 1024, torch.bfloat16), (32000, 3072, 384, torch.bfloat16),
                      (32000, 3000, 320, torch.float32), (32000, 3072, 512, torch.float32), (32000, 1024, 3072, torch.bfloat16),
                      (32000, 512, 3072, torch.bfloat16), (8192, 8192, 8192, torch.bfloat16)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=od, device=dev)
    ms = t(lambda: _ops.gemm_nt(a, b, bias, od, out))
    print("[persist=%s] nt %dx%dx%d %s %8.4f ms %8.1f TF" % (tag, M, N, K, "f32" if od == torch.float32 else "bf16", ms, 2.0 * M * N * K / ms / 1e9))
