"""A few launches of the default forward / backward recurrence kernels (for rocprofv3 --pmc runs)."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
whh16, whhT16 = whh.to(torch.bfloat16).contiguous(), whh.transpose(1, 2).contiguous().to(torch.bfloat16)
bhh = torch.zeros(ndir * 3 * H, device=dev)
dy = (torch.randn(T * B, H, generator=g) * 0.1).to(dev).to(torch.bfloat16)
dbi = torch.zeros(ndir * 3 * H, device=dev); dbh = torch.zeros(ndir * 3 * H, device=dev)
for _ in range(3):
    y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
    _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)
torch.cuda.synchronize()
_ops.gru_check_sync()
