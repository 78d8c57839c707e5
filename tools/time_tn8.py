"""asr_gemm_tn_acc_group_8ph (csrc/gemm8.hip) against asr_gemm_tn_acc_group: exact-integer and random checks, then alternating timings of the
model's grouped weight-gradient products: python tools/time_tn8.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16, F32 = torch.bfloat16, torch.float32


def check(K, M, N, ints=False, lda=None):
    g = torch.Generator().manual_seed(K + M + N)
    if ints:
        a = torch.randint(-4, 5, (K, lda or M), generator=g).float()
        b = torch.randint(-4, 5, (K, N), generator=g).float()
    else:
        a = torch.randn(K, lda or M, generator=g)
        b = torch.randn(K, N, generator=g)
    ad, bd = a.to(dev).to(BF16)[:, :M], b.to(dev).to(BF16)
    c0 = torch.randn(M, N, generator=g).to(dev)
    c = c0.clone()
    _ops.gemm_tn_acc_group_8ph([(ad, bd, c)])
    ref = c0.double() + ad.double().T @ bd.double()
    err = float((c.double() - ref).abs().max() / ref.abs().max())
    print("check K=%d M=%d N=%d ints=%s lda=%s: rel max err %.2e %s" % (K, M, N, ints, lda, err, "ok" if err < 2e-5 else "FAIL"))
    return err < 2e-5


def timed(fn, it=10):
    for _ in range(2):
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
for args in [(64, 256, 256, True), (96, 40, 24, True), (333, 48, 960, True), (1000, 136, 72, False), (4096, 1536, 512, False), (2048, 3000, 320, False),
             (5000, 264, 520, False, 320), (32000, 3072, 512, False)]:
    ok = check(*args) and ok
if not ok:
    print("MISMATCH")
    sys.exit(1)
T, B, H = 1000, 32, 512
for din in (512, 384):
    dgi = torch.randn(T * B, 6 * H).to(dev).to(BF16)
    x = torch.randn(T * B, din).to(dev).to(BF16)
    dgh = torch.randn(T * B, 6 * H).to(dev).to(BF16)
    h16 = torch.randn(T * B, 2 * H).to(dev).to(BF16)
    dwih = torch.zeros(6 * H, din, device=dev)
    dwhh = torch.zeros(2, 3 * H, H, device=dev)
    prods = [(dgi, x, dwih), (dgh[B:, :3 * H], h16[:-B, :H], dwhh[0]), (dgh[:-B, 3 * H:], h16[B:, H:], dwhh[1])]
    fl = 2.0 * T * B * (6 * H * din + 2 * 3 * H * H)
    res = []
    for rnd in range(3):
        res.append((timed(lambda: _ops.gemm_tn_acc_group(prods)), timed(lambda: _ops.gemm_tn_acc_group_8ph(prods))))
    print("grouped dW_ih (3072 x %d) + 2 dW_hh (1536 x 512), K = 32000: group %s us (%.0f TF)   8ph %s us (%.0f TF)" % (
        din, " ".join("%.1f" % r[0] for r in res), fl / min(r[0] for r in res) / 1e6, " ".join("%.1f" % r[1] for r in res), fl / min(r[1] for r in res) / 1e6))
    a1 = dwih.clone()
    dwih.zero_(); dwhh.zero_()
    _ops.gemm_tn_acc_group(prods)
    r1, r2 = dwih.clone(), dwhh.clone()
    dwih.zero_(); dwhh.zero_()
    _ops.gemm_tn_acc_group_8ph(prods)
    print("   8ph vs group: rel diff dW_ih %.2e dW_hh %.2e" % (float((dwih - r1).norm() / r1.norm()), float((dwhh - r2).norm() / r2.norm())))
for K, M, N in [(32000, 3000, 320), (32000, 640, 512), (32000, 640, 320), (416000, 256, 1920)]:
    a = torch.randn(K, M).to(dev).to(BF16)
    b = torch.randn(K, N).to(dev).to(BF16)
    c = torch.zeros(M, N, device=dev)
    res = []
    for rnd in range(3):
        res.append((timed(lambda: _ops.gemm_tn_acc(a, b, c)), timed(lambda: _ops.gemm_tn_acc_group_8ph([(a, b, c)]))))
    fl = 2.0 * K * M * N
    print("TN K=%d M=%d N=%d: tn_acc %s us (%.0f TF)   8ph %s us (%.0f TF)" % (K, M, N, " ".join("%.1f" % r[0] for r in res), fl / min(r[0] for r in res) / 1e6,
                                                                            " ".join("%.1f" % r[1] for r in res), fl / min(r[1] for r in res) / 1e6))
