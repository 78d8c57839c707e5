"""CPU oracle for the feature path (float64 NumPy).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

get_filterbanks / compute_logmel / compute_delta(s) restate asr/fft.py:58-99 and are pinned by tests/golden/fft.npz.
preemphasis / framesig / powspec restate python_speech_features.sigproc (absent third party, version unpinned in the
reference's README.md:17-21; call site asr/fft.py:53-55): PARITY UNPINNED, cross-checked only against numpy.fft.
update_stats / mean_and_std restate asr/data/loaders/base.py:64-80,39-41 (pinned by tests/golden/stats.npz);
augment_specgram restates asr/fft.py:21-50 (pinned by tests/golden/augment.npz under the same NumPy seed).
"""
import math

import numpy as np


def hz2mel(hz):
    return 2595 * np.log10(1 + hz / 700.)


def mel2hz(mel):
    return 700 * (10 ** (mel / 2595.0) - 1)


def get_filterbanks(nfilt=20, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    highfreq = highfreq or samplerate / 2
    melpoints = np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilt + 2)
    bins = np.floor((nfft + 1) * mel2hz(melpoints) / samplerate)
    fbank = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        lo, mid, hi = int(bins[j]), int(bins[j + 1]), int(bins[j + 2])
        for i in range(lo, mid):
            fbank[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(mid, hi):
            fbank[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fbank


def preemphasis(signal, coeff=0.97):
    signal = np.asarray(signal, dtype=np.float64)
    return np.append(signal[0], signal[1:] - coeff * signal[:-1])


def framesig(sig, frame_len, frame_step, winfunc=lambda n: np.ones((n,))):
    slen = len(sig)
    frame_len, frame_step = int(round(frame_len)), int(round(frame_step))
    numframes = 1 if slen <= frame_len else 1 + int(math.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    padsignal = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = np.arange(frame_len)[None, :] + (np.arange(numframes) * frame_step)[:, None]
    return padsignal[idx] * winfunc(frame_len)[None, :]


def powspec(frames, nfft):
    return 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))


def get_specgram(signal, samplerate=16000, winlen=0.025, winstep=0.01, nfft=512, preemph=0.97, winfunc=lambda n: np.ones((n,))):
    """asr/fft.py:52-56"""
    return powspec(framesig(preemphasis(signal, preemph), winlen * samplerate, winstep * samplerate, winfunc), nfft)


def compute_logmel(pspec, fbank):
    feat = np.dot(pspec, fbank.T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    return np.log(feat)


def compute_delta(feat, N=1):
    denominator = 2 * sum(i ** 2 for i in range(1, N + 1))
    padded = np.pad(feat, ((N, N), (0, 0)), mode="edge")
    out = np.empty_like(feat)
    for t in range(len(feat)):
        out[t] = np.dot(np.arange(-N, N + 1), padded[t:t + 2 * N + 1]) / denominator
    return out


def compute_deltas(logmel):
    delta = compute_delta(logmel, 1)
    delta_delta = compute_delta(delta, 1)
    return logmel[:-2], delta[:-2], delta_delta[:-2]


def logfbank_minibatch(signals, frame_len=512, frame_step=160, nfft=512, nfilt=40, samplerate=16000, mean=None, std=None):
    """Processor.extract_batch_features + features_to_minibatch (asr/data/processing.py:67-173), Hann window."""
    fbank = get_filterbanks(nfilt, nfft, samplerate)
    feats = []
    for s in signals:
        ps = powspec(framesig(preemphasis(s), frame_len, frame_step, np.hanning), nfft)
        feats.append([f.T for f in compute_deltas(compute_logmel(ps, fbank))])
    Tmax = max(f[0].shape[1] for f in feats)
    x = np.zeros((len(signals), 3, nfilt, Tmax), dtype=np.float32)
    lens = []
    for b, f in enumerate(feats):
        T = f[0].shape[1]
        for c in range(3):
            x[b, c, :, :T] = f[c]
        lens.append(T)
    if mean is not None:
        x = (x - mean[None, ..., None]) / std[None, ..., None]
        for b, T in enumerate(lens):      # the reference normalises the zero padding too (asr/data/loaders/base.py:24)
            pass
    return x, np.asarray(lens, dtype=np.int32)


class RunningStats(object):
    """asr/data/loaders/base.py:64-80 (_update_stats_recursively) and :39-41 (get_mean_and_std)."""

    def __init__(self):
        self.total, self.mean, self.nvar = 0, None, None

    def update(self, x):                    # x (3, nmel, n)
        n = x.shape[2]
        if self.total == 0:
            self.mean = np.mean(x, axis=2)
            self.nvar = np.var(x, axis=2) * n
        else:
            s, s2 = np.sum(x, axis=2), np.sum(x ** 2, axis=2)
            new_mean = self.mean + (s - n * self.mean) / (self.total + n)
            self.nvar = self.nvar + s2 - s * (new_mean + self.mean) + n * new_mean * self.mean
            self.mean = new_mean
        self.total += n

    def mean_and_std(self):
        return self.mean[None, ..., None], np.sqrt(self.nvar[None, ..., None] / (self.total - 1))


def augment_specgram(pspec, change_speech_rate=True, change_vocal_tract=True, rng=np.random):
    """asr/fft.py:21-50 with the reference's draw order (speed first, then vocal-tract ratio)."""
    new_pspec = None
    if change_speech_rate:
        speed = max(min(rng.normal(1, 0.15), 1.2), 0.8)
        new_length = int(len(pspec) / speed)
        idx = (np.arange(new_length) * speed).astype(int)
        new_pspec = pspec[idx].astype(np.float64)
        pspec = new_pspec
    if change_vocal_tract:
        dim = pspec.shape[1]
        ratio = max(min(rng.normal(1, 0.15), 1.2), 0.8)
        src = (np.arange(dim) * ratio).astype(int)
        src = np.where(src < dim, src, dim - 1)
        pspec = pspec[:, src]
    return pspec
