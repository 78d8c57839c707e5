"""Time the GRU recurrence kernels alone (one layer, both directions) in both launch modes."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch
from asr import _ops

def main(T=1000, B=32, H=512, ndir=2, iters=3):
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
    whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
    whh16 = whh.to(torch.bfloat16).contiguous()
    whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
    bhh = torch.zeros(ndir * 3 * H, device=dev)
    dy = torch.randn(T * B, H, generator=g).to(dev).to(torch.bfloat16)
    dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
    ref = None
    fref = None
    for mode in (1, 2, 4, 7, 8, 9):
        _ops.GRU_MODE[0] = mode
        res = {}
        for name in ("fwd", "bwd"):
            y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
            fn = (lambda: _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)) if name == "fwd" else \
                 (lambda: _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh))
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): fn()
            e1.record(); torch.cuda.synchronize()
            res[name + "_us_per_step"] = e0.elapsed_time(e1) / iters / T * 1e3
        _ops.gru_check_sync()
        out = _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh); torch.cuda.synchronize()
        _ops.gru_check_sync()
        fo = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir); torch.cuda.synchronize(); _ops.gru_check_sync()
        if fref is None: fref = [o.float().clone() for o in fo]
        else: res['fwd_maxdiff'] = max(float((o.float() - r).abs().max()) for o, r in zip(fo, fref))
        if ref is None: ref = [o.float().clone() for o in out[:2]]
        else: res['maxdiff_vs_step'] = max(float((o.float() - r).abs().max()) for o, r in zip(out[:2], ref))
        if mode == 8 and _ops.gru_gi_dtype(T, B, H, ndir) == torch.bfloat16:       # the default form with the input projections in bf16
            gi16 = gi.to(torch.bfloat16)
            fn = lambda: _ops.gru_fwd(gi16, whh16, bhh, T, B, H, ndir)
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters): fn()
            e1.record(); torch.cuda.synchronize()
            res["fwd_bf16gi_us_per_step"] = e0.elapsed_time(e1) / iters / T * 1e3
        print(json.dumps(dict(mode=mode, T=T, B=B, H=H, **res)))
    _ops.GRU_MODE[0] = 0

if __name__ == "__main__":
    main()
    main(B=16)
    main(B=8)
