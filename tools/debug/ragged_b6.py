import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.set_num_threads(16)
import test_gru_lengths_gpu as tg
from asr import _ops
dev = torch.device("cuda:0")
for (T, B, I, H) in [(150, 6, 128, 512), (150, 8, 128, 512), (150, 4, 128, 512), (150, 6, 128, 256), (150, 32, 128, 512)]:
    g = torch.Generator().manual_seed(B * T)
    for lens in ("ragged", "full"):
        x_len = torch.randint(T // 3, T + 1, (B,), generator=g, dtype=torch.int32) if lens == "ragged" else None
        for ps in (32, None):
            got, f32, m = tg._layer_and_reference(dev, T, B, I, H, 2, seed=5, x_len=x_len, ps_units=ps)
            print((T, B, I, H), lens, "ps", ps, "gi", _ops.gru_gi_dtype(T, B, H, 2), {n: "%.1e/%.1e" % (tg._rel(got[n], f32[n]), tg._rel(got[n], m[n])) for n in got})
