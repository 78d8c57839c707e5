"""One NT GEMM shape, a few launches (for rocprofv3 counter passes): python tools/nt_one.py M N K [f32|bf16]"""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
M, N, K = (int(v) for v in sys.argv[1:4])
od = torch.float32 if (len(sys.argv) > 4 and sys.argv[4] == "f32") else torch.bfloat16
dev = torch.device("cuda:0")
a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=od, device=dev)
for _ in range(5):
    _ops.gemm_nt(a, b, None, od, out)
torch.cuda.synchronize()
