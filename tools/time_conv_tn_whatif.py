"""What-if timings of the convolution weight-gradient kernel on the two conv blocks of the BASELINE model (invalid results)."""
import sys, os, torch
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
    return e0.elapsed_time(e1) / n
T, B = 1000, 32
for (Hin, Ci, Co, ph, name) in [(40, 8, 128, 0, "conv1 (3 -> 8 padded channels)"), (13, 64, 128, 0, "conv2")]:
    KH, KW = 3, 5
    x = torch.randn(T, B, Hin, Ci, device=dev).to(torch.bfloat16)
    Hout = Hin + 2 * ph - KH + 1
    gy = torch.randn(T, B, Hout, Co, device=dev).to(torch.bfloat16)
    sc = torch.zeros(Co, KH * KW * Ci, device=dev)
    ms = t(lambda: _ops.conv_tn_acc(gy.reshape(-1, Co), x, sc, KH, KW, ph, KW - 1, T, Hout))
    if _ops.conv_tn_copies(Co, Ci, KH, KW) == 8:
        sc8 = torch.zeros(8, Co, KH * KW * Ci, device=dev)
        ms8 = t(lambda: _ops.conv_tn_acc(gy.reshape(-1, Co), x, sc8, KH, KW, ph, KW - 1, T, Hout))
        print("   one copy per XCD: %.1f us" % (ms8 * 1e3))
    print("[whatif=%s] %s: %.1f us (gy %.0f MB)" % (os.environ.get("ASR_TN_WHATIF", "0"), name, ms * 1e3, gy.numel() * 2 / 1e6))
