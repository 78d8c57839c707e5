"""first-poll delay sweep of the forward recurrence (ASR_DEBUG gru_poll_delay is read once per process: one process per value)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, torch
sys.path.insert(0, os.path.join(%r, "chainer-speech-recognition_amd"))
from asr import _ops
T, B, H, ndir = 1000, 32, 512, 2
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(torch.bfloat16)
whh16 = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev).to(torch.bfloat16).contiguous()
bhh = torch.zeros(ndir * 3 * H, device=dev)
fn = lambda: _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
for _ in range(3): fn()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) / 10 / T * 1e3)
_ops.gru_check_sync()
print(json.dumps(dict(delay=os.environ.get("ASR_DEBUG"), fwd_us_per_step=best)))
''' % ROOT
for d in sys.argv[1:] or ["2", "3", "4", "5", "6", "7", "8", "10"]:
    env = dict(os.environ, ASR_DEBUG="gru_poll_delay=%s" % d)
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:])
