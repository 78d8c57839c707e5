"""Per-phase cycles of the forward recurrence's role-specialised loop (needs a library built with -DASR_STAMP_DP: ASR_HIP_LIB=...)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(torch.bfloat16)
whh16 = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev).to(torch.bfloat16).contiguous()
bhh = torch.zeros(ndir * 3 * H, device=dev)
for _ in range(2):
    _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
torch.cuda.synchronize()
s = _ops.LAST_SYNC[0].cpu().view(torch.uint8)
st = s[4096:4096 + 2 * 4 * 8 * 8].view(torch.int64).reshape(2, 4, 8)
names = ["top (gi)", "poll", "mfma->partials", "barrier", "gate math", "store+sleep+fetch"]
for wg in (0, 1):
    for w in range(4):
        print("wg", wg, "wave", w, " ".join("%s=%d" % (n, st[wg, w, i].item() // T) for i, n in enumerate(names)),
              "sum=%d" % (st[wg, w, :6].sum().item() // T), "attempts/step=%.2f" % (st[wg, w, 7].item() / T))
io = s[4096 + 64 * 8:4096 + 64 * 8 + 2 * 2 * 8 * 8].view(torch.int64).reshape(2, 2, 8)
for wg in (0, 1):
    for i, nm in enumerate(("loader", "storer")):
        print("wg", wg, nm, "barrier..next barrier arrival=%d" % (io[wg, i, 2].item() // T), "wait in barrier=%d" % (io[wg, i, 3].item() // T))

