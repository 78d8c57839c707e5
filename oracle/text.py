"""TEST INFRASTRUCTURE -- CPU restatement of the reference's text side: tokeniser, minibatch label assembly, greedy
collapse and character error rate.  Pinned by tests/golden/text.json and tests/golden/minibatch.npz (outputs of the
reference's own asr/vocab.py, asr/error.py and Processor.features_to_minibatch, see tests/golden/make_golden.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import numpy as np


def tokenize(sentence, sutegana, collapse):
    """asr/vocab.py:107-126"""
    def join(chars):
        out = []
        for ch in chars:
            if ch in sutegana:
                out[-1] += ch
            else:
                out.append(ch)
        return out
    first = [collapse.get(t, t) for t in join(sentence)]
    return join(ch for t in first for ch in t)


def unigram_ids(unigram_tokens):
    """asr/vocab.py:63-76"""
    ids = {"_": 0}
    for t in unigram_tokens:
        ids[t] = len(ids)
    return ids


def collapse_greedy(frame_ids, blank):
    """asr/error.py:38-47"""
    out, prev = [], blank
    for i in frame_ids:
        i = int(i)
        if i == blank:
            prev = blank
            continue
        if i == prev:
            continue
        out.append(i)
        prev = i
    return out


def levenshtein(r, h):
    """asr/error.py:9-23 (exact integers; the reference's uint8 table agrees up to 255 tokens)"""
    d = np.zeros((len(r) + 1, len(h) + 1), dtype=np.int64)
    d[0, :] = np.arange(len(h) + 1)
    d[:, 0] = np.arange(len(r) + 1)
    for i in range(1, len(r) + 1):
        for j in range(1, len(h) + 1):
            d[i, j] = d[i - 1, j - 1] if r[i - 1] == h[j - 1] else 1 + min(d[i - 1, j - 1], d[i, j - 1], d[i - 1, j])
    return int(d[len(r), len(h)])


def character_error_rate(r, h):
    """asr/error.py:7-24"""
    if len(r) == 0:
        return len(h)
    return float(levenshtein(r, h)) / len(r)


def minibatch_error(y_batch, t_batch, blank):
    """asr/error.py:26-68 for an inventory whose tokens re-tokenise to themselves"""
    total = 0.0
    for y, t in zip(y_batch, t_batch):
        target = [int(i) for i in t if int(i) != blank]
        total += character_error_rate(target, collapse_greedy(y, blank))
    return total / len(y_batch)


def labels_for_ctc(sentence, x_length, token_ids, sutegana, collapse):
    """asr/data/processing.py:131-166: unigram ids, bigram ids (-1 = out of vocabulary, first position -1), cut to what
    x_length frames can align (2L + 1 + repeats, repeats counted cyclically through np.roll)."""
    toks = tokenize(sentence, sutegana, collapse)
    uni = [token_ids[t] for t in toks]
    big = [-1] + [token_ids.get(a + b, -1) for a, b in zip(toks[:-1], toks[1:])]
    rep = int(np.count_nonzero(np.asarray(uni) == np.roll(np.asarray(uni), 1))) if uni else 0
    if x_length < 2 * len(uni) + 1 + rep:
        keep = (x_length - rep - 1) // 2
        uni, big = uni[:keep], big[:keep]
    return uni, big


def cmn_power_spectrum(pspec):
    """asr/data/processing.py:86-89 (inline statement, no separate reference function to call): parity unpinned"""
    lg = np.log(pspec)
    return np.exp(lg - lg.mean(axis=0))
