"""Time the CTC sweep (C ABI, fused loss+grad) at the BASELINE shape with HIP events on the launch stream."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chainer-speech-recognition_amd"))
import torch
from asr import _lib

def main(T=1000, B=32, V=3000, L=120, gram=0, iters=20):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    xs = torch.randn(T, B, V, generator=g).to(dev)
    lab = torch.randint(1, V if not gram else 119, (B, L), generator=g, dtype=torch.int32).to(dev)
    big = None
    if gram:
        big = torch.randint(119, V, (B, L), generator=g, dtype=torch.int32)
        big[torch.rand(B, L, generator=g) < 0.3] = -1
        big[:, 0] = -1
        big = big.to(dev)
    tl = torch.randint(40, L + 1, (B,), generator=g, dtype=torch.int32).to(dev)
    lib = _lib.lib()
    n = lib.asr_ctc_workspace_bytes(T, B, V, L, gram)
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    loss = torch.empty(B, device=dev); lm = torch.empty((), device=dev); grad = torch.empty_like(xs)
    s = torch.cuda.current_stream()
    def run():
        rc = lib.asr_ctc_loss_grad(s.cuda_stream, xs.data_ptr(), lab.data_ptr(), big.data_ptr() if gram else None, None, tl.data_ptr(),
                                   T, B, V, L, 0, 1.0 / B, loss.data_ptr(), lm.data_ptr(), grad.data_ptr(), ws.data_ptr(), n)
        assert rc == 0, rc
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s)
    for _ in range(iters): run()
    e1.record(s); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    alg = 2.0 * T * B * V * 4
    print(json.dumps(dict(gram=gram, T=T, B=B, V=V, L=L, ms=ms, alg_GBps=alg / ms / 1e6, loss_mean=lm.item())))

if __name__ == "__main__":
    main(gram=0)
    main(gram=1)
