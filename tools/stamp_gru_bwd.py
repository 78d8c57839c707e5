"""Per-phase time of the backward persistent I/O kernel (needs a library built with -DASR_STAMP)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
mode = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
whh16 = whh.to(torch.bfloat16).contiguous()
whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
bhh = torch.zeros(ndir * 3 * H, device=dev)
dy = torch.randn(T * B, H, generator=g).to(dev).to(torch.bfloat16)
dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
_ops.GRU_MODE[0] = mode
y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
for _ in range(2):
    _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)
torch.cuda.synchronize()
s = _ops.LAST_SYNC[0].cpu().view(torch.uint8)
st = s[4096:4096 + 8 * 6 * 8 * 8].view(torch.int64).reshape(8, 6, 8)
names = ["poll", "bar1", "ld+mfma", "bar2", "reduce+gates|io", "drain", "bar3", "-"]
for wg in (0, 1):
    for w in range(6):
        print("wg", wg, "wave", w, " ".join("%s=%.2f" % (n, st[wg, w, i].item() / 100.0 / T) for i, n in enumerate(names[:7])),
              "sum=%.2f" % (st[wg, w].sum().item() / 100.0 / T))

w32 = _ops.LAST_SYNC[0].cpu().view(torch.uint8)[:4096].view(torch.int32)
print("placement words: xcc", w32[960:968].tolist(), "mismatch", w32[976:984].tolist(), "arrivals", w32[992:1000].tolist())
