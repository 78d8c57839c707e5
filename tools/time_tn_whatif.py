"""What-if timings of the TN (weight-gradient) shapes (invalid results; needs a library with the ASR_TN_WHATIF knob)."""
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
for (K, M, N) in [(32000, 3072, 512), (32000, 3072, 1024), (32000, 1536, 512), (32000, 3000, 320)]:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    ms = t(lambda: _ops.gemm_tn_acc(a, b, c))
    print("[whatif=%s] tn K%d %dx%d %8.4f ms %8.1f TF-equivalent" % (os.environ.get("ASR_TN_WHATIF", "0"), K, M, N, ms, 2.0 * M * N * K / ms / 1e9))
    if os.environ.get("ASR_TN_WHATIF", "0") == "0":
        ms = t(lambda: torch.mm(a.T, b))
        print("   torch.mm(a.T, b) %8.4f ms %8.1f TF" % (ms, 2.0 * M * N * K / ms / 1e9))
