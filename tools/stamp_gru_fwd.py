"""Per-phase cycles of the forward persistent I/O kernel (needs a library built with -DASR_STAMP); us at 2.4 GHz."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
whh16 = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev).to(torch.bfloat16).contiguous()
bhh = torch.zeros(ndir * 3 * H, device=dev)
_ops.GRU_MODE[0] = mode
for _ in range(2):
    _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
torch.cuda.synchronize()
s = _ops.LAST_SYNC[0].cpu().view(torch.uint8)
st = s[4096:4096 + 8 * 6 * 12 * 8].view(torch.int64).reshape(8, 6, 12)
names = ["poll", "bar1", "loads", "mfma+lds", "bar2", "gates", "stores", "drain", "bar3", "reduce", "gi_lds", "-"]
att = st[:, :, 11].clone(); st[:, :, 11] = 0
for wg in (0, 3):
    for w in range(6):
        print("wg", wg, "wave", w, " ".join("%s=%.2f" % (n, st[wg, w, i].item() / 2400.0 / T) for i, n in enumerate(names[:11])),
              "sum=%.2f" % (st[wg, w].sum().item() / 2400.0 / T), "attempts/step=%.2f" % (att[wg, w].item() / T))

w32 = _ops.LAST_SYNC[0].cpu().view(torch.uint8)[:4096].view(torch.int32)
print("placement words: xcc", w32[960:968].tolist(), "mismatch", w32[976:984].tolist(), "arrivals", w32[992:1000].tolist())
