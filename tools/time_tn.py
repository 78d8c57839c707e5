"""TN (weight-gradient) GEMMs of the train step alone: dense shapes and the implicit convolution form."""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (K, M, N) in [(32000, 3072, 1024), (32000, 3072, 640), (32000, 1536, 512), (32000, 640, 512), (32000, 3000, 320), (32000, 320, 1024)]:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    ms = t(lambda: _ops.gemm_tn_acc(a, b, c))
    print(json.dumps(dict(kind="tn", K=K, M=M, N=N, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1))))
for (T, B, H, Cs, Co, KH, KW, ph) in [(1000, 32, 20, 64, 128, 3, 5, 1), (1000, 32, 40, 8, 128, 3, 5, 1)]:
    x = torch.randn(T, B, H, Cs, device=dev).to(torch.bfloat16)
    g = torch.randn(T * B * H, Co, device=dev).to(torch.bfloat16)
    sc = torch.zeros(Co, KH * KW * Cs, device=dev)
    ms = t(lambda: _ops.conv_tn_acc(g, x, sc, KH, KW, ph, KW - 1, T, H))
    print(json.dumps(dict(kind="conv_tn", rows=T * B * H, Co=Co, K=KH * KW * Cs, ms=round(ms, 4), tflops=round(2.0 * T * B * H * Co * KH * KW * Cs / ms / 1e9, 1))))
