"""A few launches of the fused first block (csrc/conv_first.hip) at BASELINE size, for rocprofv3 passes.  python tools/conv_first_once.py [Co]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch
from asr import _ops
BF16 = _ops.BF16
dev = torch.device("cuda:0")
Co = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T, B, Hin, Ci, KH, KW, k = 1000, 32, 40, 3, 3, 5, 3
pt, ph, Hout = KW - 1, 0, Hin - KH + 1
torch.manual_seed(0)
x = torch.randn(B, Ci, Hin, T, device=dev)
W = torch.randn(Co, Ci, KH, KW, device=dev) * 0.3
bias = torch.randn(Co, device=dev)
x8 = _ops.pack_input_pad(x, (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), T, B, Hin, Ci, 8)
Wp = torch.zeros(Co, 8, KH, KW, device=dev); Wp[:, :Ci] = W
w128 = _ops.conv_weight_pack(Wp, Kp=128)
gW, gb = torch.zeros(Co, Ci, KH, KW, device=dev), torch.zeros(Co, device=dev)
for _ in range(5):
    y, idx = _ops.conv_mp_fwd(x8, w128, bias, KH, KW, ph, pt, T, Hout, k)
    gy = y
    _ops.conv_mp_bwd(gy, idx, x8, gW, gb, KH, KW, ph, pt, Hout, k)
torch.cuda.synchronize()
print("done")
