"""eight-wave TN kernel on a GRU layer's grouped weight gradients (dW_ih 3072 x 512 + 2 dW_hh 1536 x 512) for a sweep of K = T B: the fixed
cost of a work item (fill, 256 x 256 float atomics, turnover) against its K loop.  ASR_DEBUG tn8_whatif=1 (one atomic per wave, results
invalid) and tn8_items=N (work items aimed at) are read once per process: run it once per setting."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
B, H, din = 32, 512, 512
print("ASR_DEBUG=%s" % os.environ.get("ASR_DEBUG", ""))
for T in (125, 250, 500, 1000, 2000, 4000):
    dgi = torch.randn(T * B, 6 * H).to(dev).to(BF16)
    x = torch.randn(T * B, din).to(dev).to(BF16)
    dgh = torch.randn(T * B, 6 * H).to(dev).to(BF16)
    h16 = torch.randn(T * B, 2 * H).to(dev).to(BF16)
    dwih = torch.zeros(6 * H, din, device=dev)
    dwhh = torch.zeros(2, 3 * H, H, device=dev)
    prods = [(dgi, x, dwih), (dgh[B:, :3 * H], h16[:-B, :H], dwhh[0]), (dgh[:-B, 3 * H:], h16[B:, H:], dwhh[1])]
    fl = 2.0 * T * B * (6 * H * din + 2 * 3 * H * H)
    for _ in range(5):
        _ops.gemm_tn_acc_group_8ph(prods)
    torch.cuda.synchronize()
    best = 1e9
    for rnd in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            _ops.gemm_tn_acc_group_8ph(prods)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
    print("K=%6d: %.1f us  (%.0f TFLOP/s)" % (T * B, best, fl / best / 1e6))
