"""What-if timings of one NT shape (invalid results; needs a library built with the ASR_NT_WHATIF knob): M N K"""
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
for (M, N, K) in [(32000, 3072, 512), (32000, 3072, 1024), (8192, 8192, 8192)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    ms = t(lambda: _ops.gemm_nt(a, b, None, torch.bfloat16, out))
    print("[whatif=%s] nt %dx%dx%d %8.4f ms %8.1f TF-equivalent" % (os.environ.get("ASR_NT_WHATIF", "0"), M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
    ms = t(lambda: torch.mm(a, b.T, out=out))
    print("   torch.mm %8.4f ms %8.1f TF" % (ms, 2.0 * M * N * K / ms / 1e9))
