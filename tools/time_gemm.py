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
for (M, N, K, od) in [(32000, 3072, 512, torch.float32), (32000, 3072, 512, torch.bfloat16), (32000, 3072, 2048, torch.bfloat16),
                      (32000, 512, 3072, torch.bfloat16), (32000, 640, 512, torch.bfloat16), (32000, 3000, 320, torch.float32),
                      (352000, 128, 960, torch.bfloat16), (8192, 8192, 8192, torch.bfloat16)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=od, device=dev)
    ms = t(lambda: _ops.gemm_nt(a, b, None, od, out))
    print(json.dumps(dict(kind="nt", M=M, N=N, K=K, out=str(od), ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1))))
for (K, M, N) in [(32000, 3072, 512), (32000, 1536, 512), (32000, 640, 512), (32000, 3000, 320), (352000, 128, 960)]:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    ms = t(lambda: _ops.gemm_tn_acc(a, b, c))
    print(json.dumps(dict(kind="tn", K=K, M=M, N=N, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1))))
