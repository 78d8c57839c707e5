"""Direct (LDS-resident) convolution against the implicit-GEMM kernels on the shapes of the BASELINE model and of the recipes."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
T, B = 1000, 32
for name, Hin, Ci, Co, KH, KW, ph in (("conv2 of configs[1] (64 -> 128, H 13 -> 11)", 13, 64, 128, 3, 5, 0), ("recipe conv (128 -> 256, H 13, pad 1)", 13, 128, 256, 3, 5, 1),
                                      ("recipe conv (128 -> 512, H 13, pad 1)", 13, 128, 512, 3, 5, 1), ("64 -> 64, H 13", 13, 64, 64, 3, 5, 1), ("32 -> 64, H 38", 38, 32, 64, 3, 5, 0),
                                      ("256 -> 256, H 13, pad 1", 13, 256, 256, 3, 5, 1)):
    pt = KW - 1
    Hout = Hin + 2 * ph - KH + 1
    x = torch.randn(T, B, Hin, Ci, device=dev).to(torch.bfloat16)
    W = (torch.randn(Co, Ci, KH, KW, device=dev) * 0.05)
    bias = torch.randn(Co, device=dev)
    w16 = _ops.conv_weight_pack(W)
    gy = torch.randn(T, B, Hout, Co, device=dev).to(torch.bfloat16)
    wb = _ops.conv_weight_pack_bwd(W)
    flop = 2.0 * T * B * Hout * Co * KH * KW * Ci
    flop_b = 2.0 * T * B * Hin * Ci * KH * KW * Co
    def both(xin, w, bs, sgn, Tr, Hr):
        a32 = t(lambda: _ops.conv_nt(xin, w, bs, torch.float32, KH, KW, ph, pt, sgn, Tr, Hr))
        try:
            dd = t(lambda: _ops.conv_direct_nt(xin, w, bs, KH, KW, ph, pt, sgn, Tr, Hr))
        except Exception:
            dd = float("nan")
        return a32, dd
    a, d = both(x, w16, bias, +1, T, Hout)
    ab, db = both(gy, wb, None, -1, T, Hin)
    a16 = ab16 = float("nan")
    print("%s: forward implicit f32 %.0f us, asr_conv_nt bf16 %.0f us, direct %.0f us (%.0f TFLOP/s); backward-data implicit f32 %.0f, asr_conv_nt bf16 %.0f, direct %.0f us (%.0f TFLOP/s)"
          % (name, a, a16, d, flop / d / 1e6, ab, ab16, db, flop_b / db / 1e6), flush=True)
