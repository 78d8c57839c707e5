"""asr/data/processing.py:44-173 with the same class and method names, the arithmetic on the GPU.

``extract_batch_features`` returns, in place of the reference's list of per-utterance NumPy triples, a ``DeviceFeatures``
object: the zero-padded minibatch (B, 3, nmel, Tmax) already on the GPU plus the frame counts.  It can be indexed like
the reference's list (``features[i]`` -> (logmel, delta, delta_delta) views of shape (nmel, T_i)); ``features_to_minibatch``
uses the padded array as it is.
"""
import numpy as np
import torch

from .. import _ops
from .. import fft
from ..vocab import convert_sentence_to_unigram_tokens


class DeviceFeatures(object):
    def __init__(self, x, lengths):
        self.x = x                      # (B, 3, nmel, Tmax) f32, zero beyond each utterance
        self.lengths = lengths          # list of int

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        n = self.lengths[i]
        return tuple(self.x[i, c, :, :n] for c in range(self.x.shape[1]))


def truncate_labels_for_ctc(unigram_ids, bigram_ids, x_length):
    """asr/data/processing.py:158-166: a transcription of L tokens with R repeats needs 2L + 1 + R frames; if the
    utterance is shorter the labels are cut to (x_length - R - 1) // 2 tokens.  R counts ids equal to their cyclic
    predecessor (np.roll: the first token is compared with the last), as the reference does."""
    ids = np.asarray(unigram_ids)
    repeats = int(np.count_nonzero(ids == np.roll(ids, 1))) if len(ids) else 0
    if x_length < len(unigram_ids) * 2 + 1 + repeats:
        keep = (x_length - repeats - 1) // 2
        unigram_ids, bigram_ids = unigram_ids[:keep], bigram_ids[:keep]     # a negative `keep` slices from the end, as there
    return unigram_ids, bigram_ids


class Processor(fft.Processor):

    def extract_batch_features(self, batch, augmentation=None, apply_cmn=False):
        """batch: list of (signal int16 1-d array, sentence).  White-noise augmentation (asr/data/processing.py:74-78) and
        the speed / vocal-tract warps (:83-84, asr/fft.py:21-50; random factors drawn on the host in the reference's order)
        and CMN in the log-power domain (:86-89) all run on the GPU."""
        signals = [np.asarray(s) for s, _ in batch]
        sentences = [sent for _, sent in batch]
        noise = None
        if augmentation is not None and getattr(augmentation, "add_noise", False):
            gains = np.clip(np.random.normal(200, 100, size=len(signals)), 0, 500).astype(np.float32)
            noise = (gains, int(np.random.randint(0, 2 ** 31 - 1)))
        warp = None
        if augmentation is not None and hasattr(augmentation, "using_augmentation") and augmentation.using_augmentation():
            # the reference's draw order per utterance (asr/fft.py:26,39): speed, then vocal-tract ratio
            speed, ratio = np.ones(len(signals)), np.ones(len(signals))
            for i in range(len(signals)):
                if augmentation.change_speech_rate:
                    speed[i] = max(min(np.random.normal(1, 0.15), 1.2), 0.8)
                if augmentation.change_vocal_tract:
                    ratio[i] = max(min(np.random.normal(1, 0.15), 1.2), 0.8)
            if augmentation.change_speech_rate or augmentation.change_vocal_tract:
                warp = (speed, ratio)
        x, x_length = self.logfbank_batch(signals, noise=noise, apply_cmn=apply_cmn, warp=warp)
        lengths = [int(v) for v in x_length.cpu().numpy()]
        keep = [i for i, n in enumerate(lengths) if n > 0]              # :102-103 drops empty utterances
        assert len(keep) > 0
        if len(keep) != len(lengths):
            x = x[torch.tensor(keep, device=x.device)]
            lengths = [lengths[i] for i in keep]
        sentences = [sentences[i] for i in keep]        # (the reference also forces katakana here: jaconv.hira2kata, :106)
        return DeviceFeatures(x, lengths), sentences, max(lengths), max(len(s) for _, s in batch)

    def features_to_minibatch(self, features, sentences, max_feature_length, max_sentence_length, token_ids, id_blank):
        """-> x (B, 3, nmel, Tmax) f32 on the GPU, x_length list, t (B, Lmax) int32, t_length list, bigram (B, Lmax) int32
        (host arrays, as the reference returns them before Loader.features_to_minibatch moves them)."""
        assert isinstance(token_ids, dict) and isinstance(id_blank, int)
        B = len(features)
        if isinstance(features, DeviceFeatures):
            x_batch, x_lengths = features.x[..., :max_feature_length], list(features.lengths)
        else:       # the reference's list of (logmel, delta, delta_delta) host arrays
            channels = 1 + int(self.using_delta) + int(self.using_delta_delta)
            host = np.zeros((B, channels, self.num_mel_filters, max_feature_length), dtype=np.float32)
            x_lengths = []
            for i, triple in enumerate(features):
                n = triple[0].shape[1]
                for c in range(channels):
                    host[i, c, :, :n] = np.asarray(triple[c].cpu() if isinstance(triple[c], torch.Tensor) else triple[c])
                x_lengths.append(n)
            x_batch = torch.from_numpy(host).to(self.device)
        t_batch = np.full((B, max_sentence_length), id_blank, dtype=np.int32)
        bigram_batch = np.full((B, max_sentence_length), id_blank, dtype=np.int32)
        t_lengths = []
        for i, sentence in enumerate(sentences):
            unigrams = convert_sentence_to_unigram_tokens(sentence)
            unigram_ids = [token_ids[tok] for tok in unigrams]
            bigram_ids = [-1] + [token_ids.get(a + b, -1) for a, b in zip(unigrams[:-1], unigrams[1:])]     # :131-147
            unigram_ids, bigram_ids = truncate_labels_for_ctc(unigram_ids, bigram_ids, x_lengths[i])
            n = len(unigram_ids)
            t_batch[i, :n] = unigram_ids
            bigram_batch[i, :n] = bigram_ids
            t_lengths.append(n)
        return x_batch, x_lengths, t_batch, t_lengths, bigram_batch
