"""Per-phase time of the partial-sum backward kernel (needs a library built with -DASR_STAMP: ASR_HIP_LIB=...)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(torch.bfloat16)
whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
whh16 = whh.to(torch.bfloat16).contiguous()
whhT16 = whh.transpose(1, 2).contiguous().to(torch.bfloat16)
bhh = torch.zeros(ndir * 3 * H, device=dev)
dy = torch.randn(T * B, H, generator=g).to(dev).to(torch.bfloat16)
dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
for _ in range(2):
    _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)
torch.cuda.synchronize()
s = _ops.LAST_SYNC[0].cpu().view(torch.uint8)
st = s[4096:4096 + 2 * 10 * 8 * 8].view(torch.int64).reshape(2, 10, 8)
# (round 4: the gate waves reduce with DPP moves, barrier R is gone -- phase 3 is empty; I/O waves: phase 6 includes their wait for the cue)
names = ["top..lds-in", "poll", "lane sums + DPP reduce", "(was barrier R)", "gate math + A image", "barrier A", "mfma+store+pause+fetch", "attempts"]
for wg in (0, 1):
    for w in (0, 2, 3, 8, 9):
        print("wg", wg, "wave", w, " ".join("%s=%.3f" % (n, st[wg, w, i].item() / 100.0 / T) for i, n in enumerate(names[:7])),
              "sum=%.3f" % (st[wg, w, :7].sum().item() / 100.0 / T), "attempts/step=%.2f" % (st[wg, w, 7].item() / T))
