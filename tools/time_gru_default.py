"""us per time step of the DEFAULT recurrence pair (one layer, both directions, bf16 input projections, half gates) for a list of
batch sizes:  python tools/time_gru_default.py [B ...]   (B > 32 runs as slabs of 32 rows: csrc/gru.hip slab_rows)"""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch
from asr import _ops


def one(T, B, H, ndir=2, iters=5, mode=0):
    dev = torch.device("cuda:0")
    _ops.GRU_MODE[0] = mode
    g = torch.Generator().manual_seed(0)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(_ops.gru_gi_dtype(T, B, H, ndir))
    whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
    whh16 = whh.to(torch.bfloat16).contiguous()
    whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
    bhh = torch.zeros(ndir * 3 * H, device=dev)
    dy = torch.randn(T * B, H, generator=g).to(dev).to(torch.bfloat16)
    dbi = torch.zeros(ndir * 3 * H, device=dev)
    dbh = torch.zeros(ndir * 3 * H, device=dev)
    y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
    res = dict(T=T, B=B, H=H, mode=mode, gi=str(gi.dtype).split(".")[-1], gates=str(gates.dtype).split(".")[-1])
    for name, fn in (("fwd", lambda: _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)),
                     ("bwd", lambda: _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh))):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        res[name + "_ms"] = round(ts[len(ts) // 2], 4)
        res[name + "_us_per_step"] = round(ts[len(ts) // 2] / T * 1e3, 4)
        res[name + "_min_max_ms"] = [round(ts[0], 4), round(ts[-1], 4)]
    _ops.gru_check_sync()
    _ops.GRU_MODE[0] = 0
    print(json.dumps(res))
    sys.stdout.flush()
    return res


if __name__ == "__main__":
    Bs = [int(a) for a in sys.argv[1:]] or [32, 48, 64, 128]
    T = int(os.environ.get("T", "1000"))
    H = int(os.environ.get("H", "512"))
    for B in Bs:
        one(T, B, H)
