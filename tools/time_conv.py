"""Implicit-GEMM convolution against im2col + GEMM (+ col2im) on the model's second conv layer."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
dev = torch.device("cuda:0")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
T, B, Hin, Ci, Co, KH, KW = 1000, 32, 13, 64, 128, 3, 5
x = torch.randn(T, B, Hin, Ci, device=dev).to(torch.bfloat16)
W = (torch.randn(Co, Ci, KH, KW, device=dev) * 0.05)
w16, w16t, wb = _ops.conv_weight_pack(W), _ops.conv_weight_pack(W, transpose=True), _ops.conv_weight_pack_bwd(W)
Hout = Hin - KH + 1
gy = torch.randn(T, B, Hout, Co, device=dev).to(torch.bfloat16)
st = (x.stride(0), x.stride(1), x.stride(2), x.stride(3))
print("fwd  im2col        %.3f ms" % t(lambda: _ops.im2col(x, st, T, B, Hin, Ci, KH, KW, 0)))
col = _ops.im2col(x, st, T, B, Hin, Ci, KH, KW, 0)
print("fwd  gemm(col)     %.3f ms" % t(lambda: _ops.gemm_nt(col, w16, None, torch.bfloat16)))
print("fwd  implicit      %.3f ms" % t(lambda: _ops.conv_nt(x, w16, None, torch.bfloat16, KH, KW, 0, KW - 1, +1, T, Hout)))
print("bwd  gemm(dcol)    %.3f ms" % t(lambda: _ops.gemm_nt(gy.reshape(-1, Co), w16t, None, torch.bfloat16)))
dcol = _ops.gemm_nt(gy.reshape(-1, Co), w16t, None, torch.bfloat16)
print("bwd  col2im        %.3f ms" % t(lambda: _ops.col2im(dcol, T, B, Hin, Ci, KH, KW, 0)))
print("bwd  implicit      %.3f ms" % t(lambda: _ops.conv_nt(gy, wb, None, torch.bfloat16, KH, KW, 0, KW - 1, -1, T, Hin)))
# first layer: (B, 3, 40, T) float32 -> 128 channels, k 3x5
B, Ci, Hin, T, Co = 32, 3, 40, 1000, 128
x = torch.randn(B, Ci, Hin, T, device=dev)
W = (torch.randn(Co, Ci, KH, KW, device=dev) * 0.05)
w16 = _ops.conv_weight_pack(W)
Wp = torch.zeros((Co, 8, KH, KW), device=dev); Wp[:, :Ci] = W
w16p = _ops.conv_weight_pack(Wp, Kp=128)
st = (x.stride(3), x.stride(0), x.stride(2), x.stride(1))
Hout = Hin - KH + 1
print("conv1 im2col       %.3f ms" % t(lambda: _ops.im2col(x, st, T, B, Hin, Ci, KH, KW, 0)))
col = _ops.im2col(x, st, T, B, Hin, Ci, KH, KW, 0)
print("conv1 gemm(col)    %.3f ms" % t(lambda: _ops.gemm_nt(col, w16, None, torch.bfloat16)))
print("conv1 pack_pad8    %.3f ms" % t(lambda: _ops.pack_input_pad(x, st, T, B, Hin, Ci, 8)))
xp = _ops.pack_input_pad(x, st, T, B, Hin, Ci, 8)
print("conv1 implicit     %.3f ms" % t(lambda: _ops.conv_nt(xp, w16p, None, torch.bfloat16, KH, KW, 0, KW - 1, +1, T, Hout)))
