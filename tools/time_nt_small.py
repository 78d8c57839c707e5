"""NT shapes with few 256 x 128 tiles (ASR_DEBUG nt_persist_min selects the threshold of the persistent kernel)."""
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
for (M, N, K) in [(32000, 384, 3072), (32000, 640, 512), (32000, 640, 320), (32000, 320, 3008), (32000, 512, 3072)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ms = t(lambda: _ops.gemm_nt(a, b, None, torch.bfloat16, out))
    print("[min=%s] nt %dx%dx%d %8.4f ms %8.1f TF" % (os.environ.get("ASR_DEBUG", "default"), M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
