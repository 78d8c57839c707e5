"""asr_gemm_nt_8ph (256 x 256 tile, eight waves, csrc/gemm8.hip) against asr_gemm_nt on the model's NT shapes: correctness on odd shapes first,
then alternating timings in one process: python tools/time_nt8.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16, F32 = torch.bfloat16, torch.float32


def check(M, N, K, out_dtype=BF16, bias=False, lda=None):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, lda or K, generator=g).to(dev).to(BF16)[:, :K]
    b = torch.randn(N, K, generator=g).to(dev).to(BF16)
    bv = torch.randn(N, generator=g).to(dev) if bias else None
    ref = a.float() @ b.float().t()
    if bv is not None:
        ref = ref + bv
    got = _ops.gemm_nt_8ph(a, b, bv, out_dtype)
    torch.cuda.synchronize()
    err = float((got.float() - ref).abs().max() / ref.abs().max())
    tol = 6e-3 if out_dtype == BF16 else 2e-5
    print("check M=%d N=%d K=%d %s bias=%s lda=%s: rel max err %.2e %s" % (M, N, K, str(out_dtype)[6:], bias, lda, err, "ok" if err < tol else "FAIL"))
    return err < tol


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


ok = True
for args in [(256, 256, 64), (256, 256, 128), (256, 256, 192), (512, 512, 512), (300, 260, 256), (1000, 384, 320), (777, 132, 448, F32, True),
             (32000, 512, 3072), (4096, 4096, 4096), (513, 1024, 1024, BF16, True, 1088), (255, 4, 64, F32), (700, 264, 72), (2000, 320, 3000, BF16, True), (300, 12, 200, F32, True)]:
    ok = check(*args) and ok
if not ok:
    print("MISMATCH")
    sys.exit(1)
for M, N, K, od in [(32000, 512, 3072, BF16), (32000, 3072, 512, BF16), (32000, 384, 3072, BF16), (32000, 3072, 384, BF16), (32000, 320, 3000, BF16), (32000, 3000, 320, F32),
                    (32000, 640, 512, BF16), (32000, 512, 640, BF16), (8192, 8192, 8192, BF16), (4096, 4096, 4096, BF16)]:
    a = torch.randn(M, K).to(dev).to(BF16)
    b = torch.randn(N, K).to(dev).to(BF16)
    out = torch.empty(M, N, dtype=od, device=dev)
    res = []
    for rnd in range(3):
        t_old = timed(lambda: _ops.gemm_nt(a, b, None, od, out))
        t_new = timed(lambda: _ops.gemm_nt_8ph(a, b, None, od, out))
        res.append((t_old, t_new))
    fl = 2.0 * M * N * K
    print("M=%d N=%d K=%d %s: gemm_nt %s us (%.0f TF)   8ph %s us (%.0f TF)" % (
        M, N, K, str(od)[6:], " ".join("%.1f" % r[0] for r in res), fl / min(r[0] for r in res) / 1e6,
        " ".join("%.1f" % r[1] for r in res), fl / min(r[1] for r in res) / 1e6))
