"""exact-integer float32 product into an output pre-filled with NaN, four launches: the check that found the buffer-store data hazard of the
persistent NT kernel (ASR8_STORE_FENCE, csrc/gemm8.hip): python tools/dbg_store_hazard.py  (ASR_DEBUG nt8pp_f32=1 for the persistent float32 form)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chainer-speech-recognition_amd"))
import torch
from asr import _ops
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(1)
M, N, K = 8000, 3072, 512
a = torch.randint(-3, 4, (M, K), generator=g).float().to(dev, torch.bfloat16)
b = torch.randint(-3, 4, (N, K), generator=g).float().to(dev, torch.bfloat16)
bias = torch.randint(-8, 9, (N,), generator=g).float().to(dev)
ref = a.float() @ b.float().T + bias
for trial in range(4):
    out = torch.full((M, N), float('nan'), device=dev)
    _ops.gemm_nt_8ph(a, b, bias, torch.float32, out)
    bad = (out != ref)
    idx = bad.nonzero()
    rows = sorted(set(idx[:, 0].tolist())); cols = sorted(set(idx[:, 1].tolist()))
    print("trial", trial, "bad", int(bad.sum()), "rows", rows[:12], len(rows), "tiles_m", sorted(set(r // 256 for r in rows)), "cols%256", sorted(set(c % 256 for c in cols)), "tiles_n", sorted(set(c // 256 for c in cols)))
    if len(idx):
        r0, c0 = idx[0].tolist()
        print("   out", out[r0, c0].item(), "ref", ref[r0, c0].item(), "nan (never written):", int(torch.isnan(out).sum()))
