"""Forward recurrence alone at the BASELINE size, default mode: python tools/time_gru_fwd.py"""
import sys, os, json, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
whh16 = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev).to(torch.bfloat16).contiguous()
bhh = torch.zeros(ndir * 3 * H, device=dev)
for tag, x in (("f32 gi", gi), ("bf16 gi", gi.to(torch.bfloat16))):
    try:
        fn = lambda: _ops.gru_fwd(x, whh16, bhh, T, B, H, ndir)
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        _ops.gru_check_sync()
        print(tag, "%.3f us per step" % (e0.elapsed_time(e1) / 5 / T * 1e3), os.environ.get("ASR_GRU_FWD_WIDE"))
    except Exception as e:
        print(tag, "failed:", str(e)[:100])
