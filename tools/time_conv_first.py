"""The first block (convolution over the 8-channel padded features -> Maxout(2) -> MaxPooling((k, 1))) at BASELINE size: the fused kernels
of csrc/conv_first.hip against the passes they replace.  python tools/time_conv_first.py [Co]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch
from asr import _ops
BF16 = _ops.BF16
dev = torch.device("cuda:0")
Co = int(sys.argv[1]) if len(sys.argv) > 1 else 128
T, B, Hin, Ci, KH, KW, k = 1000, 32, 40, 3, 3, 5, 3
pt, ph = KW - 1, 0
Hout = Hin - KH + 1


def t(fn, n=20):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.manual_seed(0)
x = torch.randn(B, Ci, Hin, T, device=dev)
W = torch.randn(Co, Ci, KH, KW, device=dev) * 0.3
bias = torch.randn(Co, device=dev)
x8 = _ops.pack_input_pad(x, (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), T, B, Hin, Ci, 8)
Wp = torch.zeros(Co, 8, KH, KW, device=dev); Wp[:, :Ci] = W
w128 = _ops.conv_weight_pack(Wp, Kp=128)
conv = _ops.conv_nt(x8, w128, bias, BF16, KH, KW, ph, pt, +1, T, Hout).reshape(T, B, Hout, Co)
y, idx = _ops.conv_mp_fwd(x8, w128, bias, KH, KW, ph, pt, T, Hout, k)
want = _ops.maxout2_pool_fwd(conv, k)
print("forward equal:", bool(torch.equal(y, want)))
gy = torch.randn(y.shape, device=dev).to(BF16)
gW, gb = torch.zeros(Co, Ci, KH, KW, device=dev), torch.zeros(Co, device=dev)
scratch = torch.zeros(Co, KH * KW * 8, device=dev)
print("us: conv_nt %.1f  maxout2_pool_fwd %.1f  | fused forward %.1f" % (
    t(lambda: _ops.conv_nt(x8, w128, bias, BF16, KH, KW, ph, pt, +1, T, Hout)), t(lambda: _ops.maxout2_pool_fwd(conv, k)),
    t(lambda: _ops.conv_mp_fwd(x8, w128, bias, KH, KW, ph, pt, T, Hout, k))))
g = _ops.maxout2_pool_bwd(conv, gy, k)
print("us: maxout2_pool_bwd (+db) %.1f  conv_tn_acc %.1f  unpack %.1f | fused backward (two kernels) %.1f" % (
    t(lambda: _ops.maxout2_pool_bwd(conv, gy, k, gb)), t(lambda: _ops.conv_tn_acc(g.reshape(-1, Co), x8, scratch, KH, KW, ph, pt, T, Hout)),
    t(lambda: _ops.conv_weight_grad_unpack(scratch, gW, 8)), t(lambda: _ops.conv_mp_bwd(gy, idx, x8, gW, gb, KH, KW, ph, pt, Hout, k))))
