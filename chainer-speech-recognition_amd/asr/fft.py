"""Log-mel filterbank features on the GPU -- drop-in for asr/fft.py plus the batched path Processor uses.

Same function names as the reference (asr/fft.py): ``get_specgram``, ``augment_specgram``, ``compute_logmel``,
``compute_deltas``, ``compute_delta``, ``get_filterbanks``, ``hz2mel``, ``mel2hz``; they take / return torch GPU tensors where the
reference used NumPy arrays.  ``logfbank_batch`` is the fused batched form of
Processor.extract_batch_features + features_to_minibatch (asr/data/processing.py:67-173) for a list of signals.
The filterbank matrix and the window are small host-side constants built once (asr/fft.py:68-82,
asr/data/processing.py:57-65).
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

F32 = torch.float32


def hz2mel(hz):
    return 2595 * np.log10(1 + hz / 700.)                      # asr/fft.py:84-85


def mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)                    # asr/fft.py:87-88


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    """The (nfilt, nfft/2+1) triangular mel filterbank of asr/fft.py:68-82 -- a CONSTANT TABLE that must equal the reference's
    bit for bit (tests/golden/fft.npz), built once on the host in float64: filter j rises over FFT bins [b_j, b_{j+1}) as
    (i - b_j) / (b_{j+1} - b_j) and falls over [b_{j+1}, b_{j+2}) as (b_{j+2} - i) / (b_{j+2} - b_{j+1}), with
    b = floor((nfft + 1) * mel2hz(linspace(hz2mel(low), hz2mel(high), nfilt + 2)) / samplerate).  Same float64 operations
    per entry as the reference's double loop, evaluated per filter on index vectors."""
    highfreq = highfreq or samplerate / 2
    assert highfreq <= samplerate / 2, "highfreq is greater than samplerate/2"
    edges = np.floor((nfft + 1) * mel2hz(np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)) / samplerate)
    fbank = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        lo, mid, hi = edges[j], edges[j + 1], edges[j + 2]
        rise = np.arange(int(lo), int(mid))
        fall = np.arange(int(mid), int(hi))
        fbank[j, rise] = (rise - lo) / (mid - lo)
        fbank[j, fall] = (hi - fall) / (hi - mid)
    return fbank


def num_frames(n_samples, frame_len, frame_step):
    """python_speech_features.sigproc.framesig: 1 + ceil((N - frame_len) / frame_step), 1 for short signals."""
    if n_samples <= frame_len:
        return 1
    return 1 + int(math.ceil((1.0 * n_samples - frame_len) / frame_step))


def _dev_const(a, device):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=F32).to(device)


def mel_bands(fbank):
    """asr_mel_bands: the sparse form of a (nfilt, nbins) mel matrix on the device (band start / padded length / offset per filter + taps),
    built once per matrix and handed to asr_specgram_bands with it"""
    table = torch.empty(int(_lib.lib().asr_mel_bands_bytes()) // 4, dtype=torch.int32, device=fbank.device)
    fb = fbank.to(F32).contiguous()
    check(_lib.lib().asr_mel_bands(stream(), ptr(fb), fb.shape[0], fb.shape[1], ptr(table), table.numel() * 4), "asr_mel_bands")
    return table


def _specgram(signals, lengths, nframes, Fmax, frame_len, frame_step, nfft, preemph, window, fbank, want_pspec, bands=None):
    dev = signals.device
    B = signals.shape[0]
    nfilt = 0 if fbank is None else fbank.shape[0]
    pspec = torch.empty((B, Fmax, nfft // 2 + 1), dtype=F32, device=dev) if want_pspec else None
    logmel = torch.empty((B, Fmax, nfilt), dtype=F32, device=dev) if fbank is not None else None
    is_f32 = 1 if signals.dtype == F32 else 0
    if not is_f32 and signals.dtype != torch.int16:
        raise TypeError("signals must be int16 or float32")
    rc = _lib.lib().asr_specgram_bands(stream(), ptr(signals), is_f32, ptr(lengths), signals.stride(0), B, frame_len, frame_step, nfft,
                                       float(preemph), ptr(window), ptr(nframes), Fmax, ptr(pspec), ptr(fbank), nfilt, ptr(logmel),
                                       ptr(bands) if fbank is not None else None)
    check(rc, "asr_specgram_bands")
    return pspec, logmel


def get_specgram(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfft=512, preemph=0.97, winfunc=lambda x: np.ones((x,))):
    """asr/fft.py:52-56: power spectrum (numframes, nfft/2+1) of one signal (1-d GPU tensor, int16 or float32)."""
    frame_len, frame_step = int(round(winlen * samplerate)), int(round(winstep * samplerate))
    dev = signal.device
    F = num_frames(signal.shape[0], frame_len, frame_step)
    lengths = torch.tensor([signal.shape[0]], dtype=torch.int32, device=dev)
    nfr = torch.tensor([F], dtype=torch.int32, device=dev)
    window = _dev_const(winfunc(frame_len), dev)
    pspec, _ = _specgram(signal.reshape(1, -1).contiguous(), lengths, nfr, F, frame_len, frame_step, nfft, preemph, window, None, True)
    return pspec[0]


def augment_specgram(pspec, change_speech_rate=True, change_vocal_tract=True):
    """asr/fft.py:21-50 with the reference's signature and the reference's random draws: the speed factor, then the vocal-tract
    ratio, each ``max(min(np.random.normal(1, 0.15), 1.2), 0.8)`` from NumPy's GLOBAL generator in that order, each drawn only when its
    switch is on -- ``np.random.seed(s)`` before the call reproduces the reference's output for the same seed (tests/golden/augment.npz).
    pspec: (F, nbins) power spectrum on the GPU; returns (int(F / speed), nbins): frame t is input frame int(t * speed), bin d is input
    bin int(d * ratio) (the last bin beyond the end) -- one gather kernel (asr_augment_specgram).  The reference raises NameError for
    change_speech_rate=False with change_vocal_tract=True (``new_length`` undefined, :41); here that case keeps the length."""
    speed = max(min(np.random.normal(1, 0.15), 1.2), 0.8) if change_speech_rate == True else 1.0      # noqa: E712 (the reference's test)
    ratio = max(min(np.random.normal(1, 0.15), 1.2), 0.8) if change_vocal_tract == True else 1.0      # noqa: E712
    if change_speech_rate != True and change_vocal_tract != True:                                   # noqa: E712
        return pspec
    F = pspec.shape[0]
    new_length = int(F / speed)
    assert new_length > 0
    dev = pspec.device
    out = _ops_mod().augment_specgram(pspec.to(F32).contiguous().reshape(1, F, -1), torch.tensor([new_length], dtype=torch.int32, device=dev),
                                      torch.tensor([speed], dtype=torch.float64, device=dev),
                                      torch.tensor([ratio], dtype=torch.float64, device=dev), new_length)
    return out[0]


def compute_logmel(pspec, samplerate=16000, winlen=0.025, winstep=0.01, nfilt=26, nfft=512, lowfreq=0, highfreq=None,
                   preemph=0.97, winfunc=None, fbank=None):
    """asr/fft.py:58-66: log(pspec . fbank^T), exact zeros replaced by eps."""
    if fbank is None:
        fbank = get_filterbanks(nfilt, nfft, samplerate, lowfreq, highfreq or samplerate / 2)
    fb = fbank if isinstance(fbank, torch.Tensor) else _dev_const(fbank, pspec.device)
    pspec = pspec.to(F32).contiguous()
    out = torch.empty((pspec.shape[0], fb.shape[0]), dtype=F32, device=pspec.device)
    check(_lib.lib().asr_logmel(stream(), ptr(pspec), ptr(fb.contiguous()), pspec.shape[0], pspec.shape[1], fb.shape[0], ptr(out)),
          "asr_logmel")
    return out


def _deltas(logmel_bfm, nframes, Tmax, mean=None, std=None):
    B, Fmax, nfilt = logmel_bfm.shape
    out = torch.empty((B, 3, nfilt, Tmax), dtype=F32, device=logmel_bfm.device)
    check(_lib.lib().asr_deltas(stream(), ptr(logmel_bfm), ptr(nframes), B, Fmax, nfilt, Tmax, ptr(mean), ptr(std), ptr(out)),
          "asr_deltas")
    return out


def compute_deltas(logmel):
    """asr/fft.py:6-19: (F, nfilt) -> (logmel[:-2], delta[:-2], delta_delta[:-2])."""
    F, nfilt = logmel.shape
    nfr = torch.tensor([F], dtype=torch.int32, device=logmel.device)
    x = _deltas(logmel.to(F32).contiguous().reshape(1, F, nfilt), nfr, max(F - 2, 1))
    if F <= 2:
        x = x[..., :0]
    return x[0, 0].T, x[0, 1].T, x[0, 2].T


def compute_delta(feat, N=1):
    """asr/fft.py:90-99 for N = 1 (the only value the reference uses)."""
    if N != 1:
        raise ValueError("only N = 1 is implemented on the HIP path (asr/fft.py:8,10 use nothing else)")
    F, nfilt = feat.shape
    pad = torch.cat([feat, feat[-1:], feat[-1:]], dim=0).to(F32).contiguous()     # two extra frames are dropped again
    nfr = torch.tensor([F + 2], dtype=torch.int32, device=feat.device)
    x = _deltas(pad.reshape(1, F + 2, nfilt), nfr, F)
    return x[0, 1].T


def _ops_mod():
    from . import _ops
    return _ops


def _int16_to_f32(t):
    """exact widening of the int16 samples (a dtype conversion, no arithmetic)"""
    return t if t.dtype == F32 else t.to(F32)


class Processor(object):
    """The feature half of asr/data/processing.py:44-111 (defaults = run/ctc/cnn/args.py:18-25), batched on the GPU."""

    def __init__(self, sampling_rate=16000, frame_width=0.032, frame_shift=0.01, num_mel_filters=40, window_func="hanning",
                 using_delta=True, using_delta_delta=True, device=None):
        assert window_func in ["hanning", "hamming"]
        self.sampling_rate = sampling_rate
        self.frame_width = frame_width
        self.frame_shift = frame_shift
        self.num_fft = int(sampling_rate * frame_width)
        self.frame_len = int(round(frame_width * sampling_rate))
        self.frame_step = int(round(frame_shift * sampling_rate))
        self.num_mel_filters = num_mel_filters
        self.using_delta = using_delta
        self.using_delta_delta = using_delta_delta
        self.window = np.hanning(self.frame_len) if window_func == "hanning" else np.hamming(self.frame_len)
        self.fbank = get_filterbanks(nfft=self.num_fft, nfilt=num_mel_filters, samplerate=sampling_rate)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._window_d = self._fbank_d = self._bands_d = None

    def logfbank_batch(self, signals, mean=None, std=None, noise=None, apply_cmn=False, warp=None):
        """signals: list of 1-d int16 arrays/tensors (or a padded (B, N) tensor + lengths tuple).  Returns
        x (B, 3, nmel, Tmax) float32 on the GPU, zero padded, and x_length (B) int32 (asr/data/processing.py:113-173).
        noise = (gain per utterance, seed): white-noise augmentation (:74-78); apply_cmn: cepstral mean normalisation in
        the log-power domain (:86-89); warp = (speed per utterance, ratio per utterance) float64: asr/fft.py:21-50."""
        dev = self.device
        if self._window_d is None:
            self._window_d, self._fbank_d = _dev_const(self.window, dev), _dev_const(self.fbank, dev)
            self._bands_d = mel_bands(self._fbank_d)        # the filters' non-zero bands, once per Processor
        if isinstance(signals, tuple):
            padded, lens = signals
            lens = [int(v) for v in lens]
            padded = padded.to(dev)
        else:
            lens = [int(len(s)) for s in signals]
            host = np.zeros((len(signals), (max(lens) + 3) // 4 * 4), dtype=np.int16)      # rows of 4 n samples: the four-sample loads of asr_specgram
            for i, s in enumerate(signals):
                host[i, :lens[i]] = np.asarray(s.cpu() if isinstance(s, torch.Tensor) else s, dtype=np.int16)
            padded = torch.from_numpy(host).to(dev)
        frames = [num_frames(n, self.frame_len, self.frame_step) for n in lens]
        Fmax = max(frames)
        Tmax = max(Fmax - 2, 1)
        # the three small integer arrays of a batch travel in ONE host-to-device copy (each torch.tensor(list, device=...) is a synchronous
        # copy of its own: ~20 us apiece against 47 us for the kernels of a 32-utterance batch)
        meta = torch.tensor([lens, frames, [max(f - 2, 0) for f in frames]], dtype=torch.int32).to(dev, non_blocking=True)
        lengths, nfr = meta[0], meta[1]
        padded = padded.contiguous()
        if noise is not None:
            gains, seed = noise
            padded = _ops_mod().add_white_noise(_int16_to_f32(padded), lengths, _dev_const(gains, dev), seed)
        if apply_cmn or warp is not None:
            pspec, _ = _specgram(padded, lengths, nfr, Fmax, self.frame_len, self.frame_step, self.num_fft, 0.97,
                                 self._window_d, None, True)
            if warp is not None:
                speed, ratio = (np.asarray(w, dtype=np.float64) for w in warp)
                frames = [int(f / sp) for f, sp in zip(frames, speed)]          # new_length = int(len(specgram) / speed)
                Fmax = max(frames)
                Tmax = max(Fmax - 2, 1)
                nfr = torch.tensor(frames, dtype=torch.int32, device=dev)
                pspec = _ops_mod().augment_specgram(pspec, nfr, torch.from_numpy(speed).to(dev), torch.from_numpy(ratio).to(dev), Fmax)
            if apply_cmn:
                _ops_mod().cmn_pspec(pspec, nfr)
            logmel = compute_logmel(pspec.reshape(len(lens) * Fmax, -1), fbank=self._fbank_d).reshape(len(lens), Fmax, -1)
        else:
            _, logmel = _specgram(padded, lengths, nfr, Fmax, self.frame_len, self.frame_step, self.num_fft, 0.97,
                                  self._window_d, self._fbank_d, False, self._bands_d)
        m = None if mean is None else (mean.reshape(-1) if isinstance(mean, torch.Tensor) and mean.device == dev else _dev_const(mean, dev).reshape(-1))
        s = None if std is None else (std.reshape(-1) if isinstance(std, torch.Tensor) and std.device == dev else _dev_const(std, dev).reshape(-1))
        x = _deltas(logmel, nfr, Tmax, m, s)
        x_length = meta[2] if warp is None else torch.tensor([max(f - 2, 0) for f in frames], dtype=torch.int32, device=dev)
        return x, x_length
