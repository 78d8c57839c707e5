"""One projection-shaped NT product per launch for an L2 counter pass (rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum ...)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
M, N, K = [int(v) for v in (sys.argv[1:4] if len(sys.argv) > 3 else (32000, 3072, 512))]
a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(6): _ops.gemm_nt(a, b, None, torch.bfloat16, out)
torch.cuda.synchronize()
