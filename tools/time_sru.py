"""SRU scans (asr_sru_fwd / asr_sru_bwd through _ops) at T=1000, B=32, D in {384, 512, 1024}, tanh on / off: ms and achieved GB/s
against the algorithmic bytes  forward T B D (2 + 12 + 4 + 2),  backward T B D (2 + 12 + 4 + 2 + 6 + 2)  (x, U, C, H / gH, gU, gxh);
chunked scans against the one-thread-per-column kernels."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chainer-speech-recognition_amd"))
import torch
from asr import _ops

HBM_ACHIEVABLE = 6300.0


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    dev = torch.device("cuda:0")
    T, B = 1000, 32
    out = []
    for D in (384, 512, 1024):
        x = torch.randn(T, B, D, device=dev).to(torch.bfloat16)
        U = torch.randn(T * B, 3 * D, device=dev)
        bias = torch.randn(2 * D, device=dev) * 0.3
        c0 = torch.randn(B, D, device=dev)
        gH = torch.randn(T, B, D, device=dev).to(torch.bfloat16)
        gcT = torch.randn(B, D, device=dev)
        gb = torch.zeros(2 * D, device=dev)
        for use_tanh in (True, False):
            for chunked in (True, False):
                _ops.SRU_CHUNKED[0] = chunked
                H, C, cT = _ops.sru_fwd(x, U, bias, c0, None, use_tanh)
                f = timed(lambda: _ops.sru_fwd(x, U, bias, c0, None, use_tanh), 20 if chunked else 3)
                b = timed(lambda: _ops.sru_bwd(x, U, bias, C, c0, None, gH, gcT, gb, use_tanh), 20 if chunked else 3)
                fb, bb = T * B * D * 20.0, T * B * D * 28.0
                out.append(dict(D=D, tanh=use_tanh, chunked=chunked, fwd_ms=f, bwd_ms=b, fwd_GBps=fb / f / 1e6, bwd_GBps=bb / b / 1e6,
                                fwd_frac_of_6300=fb / f / 1e6 / HBM_ACHIEVABLE, bwd_frac_of_6300=bb / b / 1e6 / HBM_ACHIEVABLE))
                print(json.dumps(out[-1]))
    _ops.SRU_CHUNKED[0] = True


if __name__ == "__main__":
    main()
