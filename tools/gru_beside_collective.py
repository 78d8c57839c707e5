"""What does a RESIDENT collective kernel do to a persistent recurrence?  One-GPU stand-ins on a second stream while
the default forward / backward recurrence of BASELINE size (T=1000, B=32, H=512, both directions) runs:
  hold   k workgroups that keep `lds` bytes of LDS and sleep (asr_occupy_cus)     -- CUs the recurrence cannot have
  copy   k workgroups of 256 threads that stream a 256 MB buffer (asr_stream_traffic) -- an RCCL-ring-like kernel that fits BESIDE a
         recurrence workgroup on its CU (lds small) or not (lds large)
Prints us per time step of the recurrence launch (wall time of the launch / T), whether it gave up, and the stand-in's share."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch
from asr import _ops, _lib

dev = torch.device("cuda:0")
T, B, H, ndir = 1000, 32, 512, 2
g = torch.Generator().manual_seed(0)
gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(dev).to(_ops.gru_gi_dtype(T, B, H, ndir))
whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(dev)
whh16, whhT16 = whh.to(torch.bfloat16).contiguous(), whh.transpose(1, 2).contiguous().to(torch.bfloat16)
bhh = torch.zeros(ndir * 3 * H, device=dev)
dy = torch.randn(T * B, H, generator=g).to(dev).to(torch.bfloat16)
dbi, dbh = torch.zeros(ndir * 3 * H, device=dev), torch.zeros(ndir * 3 * H, device=dev)
y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
scratch = torch.zeros(64 * 1024 * 1024, device=dev)
side = torch.cuda.Stream()
lib = _lib.lib()


def run(which, kind, k, lds, hog_us=6000):
    fn = (lambda: _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)) if which == "fwd" else (lambda: _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh))
    fn()
    torch.cuda.synchronize()
    times = []
    gave_up = 0
    for _ in range(3):
        if kind == "hold":
            _lib.check(lib.asr_occupy_cus(side.cuda_stream, hog_us, lds, k), "asr_occupy_cus")
        elif kind == "copy":
            _lib.check(lib.asr_stream_traffic(side.cuda_stream, hog_us, lds, k, scratch.data_ptr(), scratch.numel() * 4), "asr_stream_traffic")
        if kind != "alone":
            lib.asr_stream_delay(_lib.stream(), 300)             # the stand-in is resident before the recurrence is queued
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
        try:
            _ops.gru_check_sync()
        except _lib.AsrHipError:
            gave_up += 1
    times.sort()
    print(json.dumps(dict(recurrence=which, beside=kind, workgroups=k, lds_kb=lds // 1024, ms=[round(t, 3) for t in times],
                          us_per_step=round(times[1] / T * 1e3, 3), gave_up=gave_up)))
    sys.stdout.flush()


for which in ("fwd", "bwd"):
    run(which, "alone", 0, 0)
    for k in (8, 16, 32):
        run(which, "copy", k, 16 * 1024)          # fits beside a recurrence workgroup (forward asks for 96 KB, backward for 132 KB of 160)
    for k in (8, 32):
        run(which, "copy", k, 120 * 1024)         # does not fit: those CUs are taken until the stand-in leaves
    run(which, "hold", 16, 120 * 1024)
