"""narrow convolutions (N <= 128) on the eight-wave kernel (asr_conv_nt_8pn) against the routed asr_conv_nt: python tools/time_conv8n.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
T, B, KH, KW = 1000, 32, 3, 5
pt = KW - 1
# (name, Hin, Ci, Co, ph, backward): DS2 conv2 (13 x 64 -> 11 x 128, no height padding) and the recipes' 128-channel layers
for name, Hin, Ci, Co, ph, bwd in [("ds2 conv2 forward   64 -> 128", 13, 64, 128, 0, False), ("ds2 conv2 backward  128 -> 64", 13, 64, 128, 0, True),
                                   ("cnn 256 -> 128 backward-data", 13, 128, 256, 1, True), ("cnn 512 -> 128 backward-data", 13, 128, 512, 1, True),
                                   ("cnn 128 -> 128 forward", 13, 128, 128, 1, False)]:
    Hout = Hin + 2 * ph - KH + 1
    W = (torch.randn(Co, Ci, KH, KW) * 0.05).to(dev)
    if bwd:
        xin = torch.randn(T, B, Hout, Co).to(dev).to(BF16)
        w2 = _ops.conv_weight_pack_bwd(W)
        args = (xin, w2, None, BF16, KH, KW, ph, pt, -1, T, Hin)
        fl = 2.0 * T * B * Hin * Ci * Co * KH * KW
    else:
        xin = torch.randn(T, B, Hin, Ci).to(dev).to(BF16)
        w2 = _ops.conv_weight_pack(W)
        args = (xin, w2, torch.zeros(Co, device=dev), BF16, KH, KW, ph, pt, +1, T, Hout)
        fl = 2.0 * T * B * Hout * Ci * Co * KH * KW
    res = {}
    for nm, fn in (("routed", _ops.conv_nt), ("8pn", _ops.conv_nt_8pn)):
        for _ in range(2):
            fn(*args)
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn(*args)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        res[nm] = best
    d = float((_ops.conv_nt(*args).float() - _ops.conv_nt_8pn(*args).float()).norm() / _ops.conv_nt(*args).float().norm())
    print("%-32s routed %.0f us (%.0f TF)   8pn %.0f us (%.0f TF)   rel diff %.1e" % (name, res["routed"], fl / res["routed"] / 1e6, res["8pn"], fl / res["8pn"] / 1e6, d))
