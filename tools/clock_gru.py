import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev)
whh16 = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev).to(torch.bfloat16).contiguous()
bhh = torch.zeros(ndir * 3 * H, device=dev)
_ops.GRU_MODE[0] = 2
for _ in range(2):
    _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
torch.cuda.synchronize()
s = _ops.LAST_SYNC[0].cpu()
v = s[256:260].view(torch.int64)
print("shader cycles", v[0].item(), "realtime ticks (100 MHz)", v[1].item(), "=> clock GHz", v[0].item() / (v[1].item() * 10.0), "us/step", v[1].item() / 100.0 / T)
