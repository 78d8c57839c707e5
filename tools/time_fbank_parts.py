"""feature kernels, parts: specgram with mel + log (the shipped path) against specgram to a power spectrum only (no band scan, no mel)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "chainer-speech-recognition_amd"))
import torch  # noqa: E402
from asr import fft  # noqa: E402

dev = torch.device("cuda", 0)
B, N = 32, 160672
proc = fft.Processor(device=dev)
sig = torch.round(torch.randn(B, N) * 3000).to(torch.int16).to(dev)
proc.logfbank_batch((sig, [N] * B))
F = fft.num_frames(N, proc.frame_len, proc.frame_step)
lengths = torch.tensor([N] * B, dtype=torch.int32, device=dev)
nfr = torch.tensor([F] * B, dtype=torch.int32, device=dev)


def timed(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / it


a = timed(lambda: fft._specgram(sig, lengths, nfr, F, proc.frame_len, proc.frame_step, proc.num_fft, 0.97, proc._window_d, proc._fbank_d, False, proc._bands_d))
b = timed(lambda: fft._specgram(sig, lengths, nfr, F, proc.frame_len, proc.frame_step, proc.num_fft, 0.97, proc._window_d, None, True))
lm = fft._specgram(sig, lengths, nfr, F, proc.frame_len, proc.frame_step, proc.num_fft, 0.97, proc._window_d, proc._fbank_d, False, proc._bands_d)[1]
c = timed(lambda: fft._deltas(lm, nfr, F - 2, None, None))
print("specgram + mel + log: %.1f us   specgram -> pspec only: %.1f us   deltas: %.1f us" % (a, b, c))
