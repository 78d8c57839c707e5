"""The step's NT products under one kernel selection (ASR_DEBUG nt_wide / nt_wide_force / nt_persist, read once per process):
python tools/time_nt_modes.py   -- prints ms and TFLOP/s per shape"""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


tag = "wide=%s force=%s persist=%s" % (os.environ.get("ASR_DEBUG", "-"), "", "")
for (M, N, K, od) in [(32000, 3072, 512, torch.bfloat16), (32000, 512, 3072, torch.bfloat16), (32000, 3072, 384, torch.bfloat16),
                      (32000, 384, 3072, torch.bfloat16), (32000, 3000, 320, torch.float32), (32000, 320, 3000, torch.bfloat16),
                      (32000, 640, 512, torch.bfloat16), (32000, 512, 640, torch.bfloat16)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    bias = torch.randn(N, device=dev)
    out = torch.empty(M, N, dtype=od, device=dev)
    ms = t(lambda: _ops.gemm_nt(a, b, bias, od, out))
    ref = (a[:256].float() @ b.float().t() + bias)
    err = float((out[:256].float() - ref).abs().max() / ref.abs().max())
    print(json.dumps(dict(sel=tag, M=M, N=N, K=K, out=str(od).split(".")[-1], ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1), err=round(err, 5))))
