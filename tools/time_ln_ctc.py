"""Backward of LayerNormalization + CTC at the BASELINE logits shape: fused sweep (csrc/ctc_ln.hip) vs ctc::grad + ln::bwd_rows."""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops, _lib
dev = torch.device("cuda:0")
T, B, V, L = 1000, 32, 3000, 120
g = torch.Generator().manual_seed(0)
x = torch.randn(T * B, V, generator=g).to(dev)
gamma, beta = torch.ones(V, device=dev), torch.zeros(V, device=dev)
y, mean, rstd = _ops.layernorm_fwd(x, gamma, beta, V, torch.float32)
lab = torch.randint(1, V, (B, L), generator=g, dtype=torch.int32).to(dev)
tl = torch.randint(40, L + 1, (B,), generator=g, dtype=torch.int32).to(dev)
lib = _lib.lib()
n = lib.asr_ctc_workspace_bytes(T, B, V, L, 0)
ws = torch.empty(n, dtype=torch.uint8, device=dev)
loss = torch.empty(B, device=dev)
s = torch.cuda.current_stream().cuda_stream
assert lib.asr_ctc_forward(s, y.data_ptr(), lab.data_ptr(), None, None, tl.data_ptr(), T, B, V, L, 0, loss.data_ptr(), None, ws.data_ptr(), n) == 0
grad = torch.empty_like(y)
dg, db = torch.zeros(V, device=dev), torch.zeros(V, device=dev)
def unfused():
    assert lib.asr_ctc_backward(s, y.data_ptr(), None, T, B, V, L, 0, None, 0, 1.0 / B, grad.data_ptr(), ws.data_ptr(), n) == 0
    return _ops.layernorm_bwd(x, grad, gamma, mean, rstd, V, torch.bfloat16, dg, db)
rec = [dict(ws=ws, Lmax=L, gram=0, x_len=None, gy=None, gy_per_utt=0, scale=1.0 / B)]
def fused():
    return _ops.layernorm_ctc_bwd(x, gamma, beta, mean, rstd, T, B, torch.bfloat16, dg, db, True, rec)
def t(fn, k=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
a, b = unfused().float(), fused().float()
print(json.dumps(dict(unfused_ms=t(unfused), fused_ms=t(fused), maxdiff=float((a - b).abs().max()), scale=float(a.abs().max()))))
