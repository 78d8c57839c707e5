"""NT GEMM / implicit convolution timings on the shapes of the two bench configurations (ASR_DEBUG nt_wide selects the kernel)."""
import sys, os, json, torch
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
mode = os.environ.get("ASR_DEBUG", "default")
res = []
for (M, N, K, od) in [(32000, 3072, 512, torch.float32), (32000, 3072, 384, torch.float32), (32000, 512, 3072, torch.bfloat16),
                      (32000, 3000, 320, torch.float32), (32000, 640, 512, torch.bfloat16), (8192, 8192, 8192, torch.bfloat16)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=od, device=dev)
    ms = t(lambda: _ops.gemm_nt(a, b, None, od, out))
    res.append(("nt %dx%dx%d %s" % (M, N, K, "f32" if od == torch.float32 else "bf16"), ms, 2.0 * M * N * K / ms / 1e9))
T, B = 1000, 32
for (Hin, Ci, Co, ph, name) in [(13, 128, 256, 1, "cnn narrow"), (13, 128, 512, 1, "cnn h->4h"), (13, 256, 512, 1, "cnn wide"), (13, 64, 128, 0, "ds2 conv2")]:
    KH, KW = 3, 5
    x = torch.randn(T, B, Hin, Ci, device=dev).to(torch.bfloat16)
    W = torch.randn(Co, Ci, KH, KW, device=dev) * 0.05
    w16, wb = _ops.conv_weight_pack(W), _ops.conv_weight_pack_bwd(W)
    Hout = Hin + 2 * ph - KH + 1
    gy = torch.randn(T, B, Hout, Co, device=dev).to(torch.bfloat16)
    fl = 2.0 * T * B * Hout * Co * Ci * KH * KW
    ms = t(lambda: _ops.conv_nt(x, w16, None, torch.bfloat16, KH, KW, ph, KW - 1, +1, T, Hout))
    res.append(("conv fwd %s" % name, ms, fl / ms / 1e9))
    ms = t(lambda: _ops.conv_nt(gy, wb, None, torch.bfloat16, KH, KW, ph, KW - 1, -1, T, Hin))
    res.append(("conv bwd-data %s" % name, ms, fl / ms / 1e9))
    sc = torch.zeros(Co, KH * KW * Ci, device=dev)
    ms = t(lambda: _ops.conv_tn_acc(gy.reshape(-1, Co), x, sc, KH, KW, ph, KW - 1, T, Hout))
    res.append(("conv bwd-weight %s" % name, ms, fl / ms / 1e9))
for (K, M, N) in [(32000, 3072, 512), (32000, 1536, 512), (32000, 3000, 320)]:
    a = torch.randn(K, M, device=dev).to(torch.bfloat16); b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.zeros(M, N, device=dev)
    ms = t(lambda: _ops.gemm_tn_acc(a, b, c))
    res.append(("tn K%d %dx%d" % (K, M, N), ms, 2.0 * M * N * K / ms / 1e9))
for name, ms, tf in res:
    print("[wide=%s] %-34s %8.4f ms %8.1f TF" % (mode, name, ms, tf))
