"""implicit convolutions of the wide recipe on the eight-wave kernel (asr_conv_nt_8ph) against the routed default with conv_8ph=0 semantics
(asr_conv_nt of the SAME process routes to 8ph, so the old kernels are timed through ASR_DEBUG conv_8ph=0 in a second process):
python tools/time_conv8.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import _ops  # noqa: E402

dev = torch.device("cuda", 0)
BF16 = torch.bfloat16
T, B, H, KH, KW = 1000, 32, 13, 3, 5
ph, pt = 1, KW - 1
for Ci, Co in [(256, 512), (128, 512), (256, 256), (512, 256)]:
    x = torch.randn(T, B, H, Ci).to(dev).to(BF16)
    W = (torch.randn(Co, Ci, KH, KW) * 0.05).to(dev)
    w16 = _ops.conv_weight_pack(W)
    bias = torch.zeros(Co, device=dev)
    fns = {"conv_nt (routed)": lambda: _ops.conv_nt(x, w16, bias, BF16, KH, KW, ph, pt, +1, T, H)}
    if os.environ.get("ASR_DEBUG", "") == "":
        fns["conv_nt_8ph"] = lambda: _ops.conv_nt_8ph(x, w16, bias, BF16, KH, KW, ph, pt, +1, T, H)
    fl = 2.0 * T * B * H * Co * Ci * KH * KW
    for name, fn in fns.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        print("%-18s %d -> %d channels forward: %.0f us (%.0f TFLOP/s)  [ASR_DEBUG=%s]" % (name, Ci, Co, best, fl / best / 1e6, os.environ.get("ASR_DEBUG", "")))
print("weight gradients (asr_conv_tn_acc routed / asr_conv_tn_acc_8ph):")
for Ci, Co in [(256, 512), (128, 256), (128, 512), (64, 128)]:
    x = torch.randn(T, B, H, Ci).to(dev).to(BF16)
    gy = torch.randn(T * B * H, Co).to(dev).to(BF16)
    acc = torch.zeros(Co, KH * KW * Ci, device=dev)
    fns = {"conv_tn_acc (routed)": lambda: _ops.conv_tn_acc(gy, x, acc, KH, KW, ph, pt, T, H)}
    if os.environ.get("ASR_DEBUG", "") == "":
        fns["conv_tn_acc_8ph"] = lambda: _ops.conv_tn_acc_8ph(gy, x, acc, KH, KW, ph, pt, T, H)
    fl = 2.0 * T * B * H * Co * Ci * KH * KW
    for name, fn in fns.items():
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        best = 1e9
        for rnd in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 5 * 1e3)
        print("%-22s %d -> %d channels: %.0f us (%.0f TFLOP/s)  [ASR_DEBUG=%s]" % (name, Ci, Co, best, fl / best / 1e6, os.environ.get("ASR_DEBUG", "")))
