import os, sys
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT)
import torch, argparse
import bench
from asr import _ops, _lib
from asr.model.architectures import build_model
from asr.loss import connectionist_temporal_classification
from asr.data.synthetic import synthetic_batch
sys.argv = [sys.argv[0]]
args = bench.parse(); args.num_conv_layers = 4
dev = torch.device("cuda:0")
cfg = bench.cnn_config(args, 119)
model = build_model(cfg).to_gpu(0)
x, labels, x_len, l_len = (t.to(dev) for t in synthetic_batch(32, 1000, 119, seed=0))
with torch.no_grad(): model(x)
orig = _ops.conv_nt
def spy(x_, W2, bias, out_dtype, KH, KW, ph, pt, sgn, Tr, Hr):
    Ts, B, Hs, Cs = x_.shape
    ok = _lib.lib().asr_conv_direct_ok(Ts, B, Hs, Cs, KH, KW, Tr, Hr, W2.shape[0], W2.shape[1], int(out_dtype == torch.bfloat16))
    print("conv_nt x", tuple(x_.shape), "N", W2.shape[0], "K", W2.shape[1], "k", (KH, KW), "pad", (ph, pt), "sgn", sgn, "Tr,Hr", (Tr, Hr), out_dtype, "direct" if ok else "IMPLICIT")
    return orig(x_, W2, bias, out_dtype, KH, KW, ph, pt, sgn, Tr, Hr)
_ops.conv_nt = spy
import asr.functions as F
loss = connectionist_temporal_classification(model(x), labels, 0, x_len, l_len)
loss.backward()
torch.cuda.synchronize()
