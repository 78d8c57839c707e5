"""CPU: oracle/text.py and the host-side mirror (asr.vocab, asr.data.processing label logic) against the reference's
own outputs in tests/golden/text.json and tests/golden/minibatch.npz."""
import json
import os

import numpy as np

from oracle import text as otext

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _text():
    with open(os.path.join(GOLD, "text.json"), encoding="utf-8") as f:
        return json.load(f)


def test_tokeniser_and_ids():
    g = _text()
    ids = otext.unigram_ids(g["unigram_tokens"])
    assert len(ids) == g["vocab_size"]
    for s, toks, tid in zip(g["sentences"], g["tokens"], g["token_ids"]):
        assert otext.tokenize(s, g["sutegana"], g["collapse"]) == toks
        assert [ids[t] for t in toks] == tid


def test_package_vocab_matches_reference():
    from asr import vocab
    g = _text()
    assert vocab.UNIGRAM_TOKENS == g["unigram_tokens"] and vocab.ID_BLANK == g["blank"]
    ids, inv = vocab.get_unigram_ids()
    assert len(ids) == g["vocab_size"] and inv[0] == "_"
    for s, toks, tid in zip(g["sentences"], g["tokens"], g["token_ids"]):
        assert vocab.convert_sentence_to_unigram_tokens(s) == toks
        assert vocab.convert_sentence_to_unigram_ids(s, ids) == tid
    assert len(vocab.get_all_bigram_tokens()) == len(g["unigram_tokens"]) ** 2


def test_cer_against_reference():
    g = _text()
    for r, h, want in zip(g["lev_r"], g["lev_h"], g["lev"]):
        assert otext.character_error_rate(r, h) == want
    y, t = np.asarray(g["y"]), np.asarray(g["t"])
    assert abs(otext.minibatch_error(y, t, g["blank"]) - g["cer_mean"]) < 1e-12
    for b in range(len(y)):
        assert abs(otext.minibatch_error(y[b:b + 1], t[b:b + 1], g["blank"]) - g["cer_each"][b]) < 1e-12


def test_minibatch_labels_and_truncation():
    g = _text()
    mb = np.load(os.path.join(GOLD, "minibatch.npz"))
    with open(os.path.join(GOLD, "minibatch.json"), encoding="utf-8") as f:
        meta = json.load(f)
    from asr.data.processing import truncate_labels_for_ctc
    from asr import vocab
    for i, s in enumerate(meta["sentences"]):
        uni, big = otext.labels_for_ctc(s, int(mb["x_length"][i]), meta["token_ids"], g["sutegana"], g["collapse"])
        n = int(mb["t_length"][i])
        assert len(uni) == n
        assert list(mb["t"][i, :n]) == uni and list(mb["bigram"][i, :n]) == big
        assert (mb["t"][i, n:] == 0).all() and (mb["bigram"][i, n:] == 0).all()
        # the product's host logic
        toks = vocab.convert_sentence_to_unigram_tokens(s)
        u = [meta["token_ids"][t] for t in toks]
        bg = [-1] + [meta["token_ids"].get(a + b, -1) for a, b in zip(toks[:-1], toks[1:])]
        u2, b2 = truncate_labels_for_ctc(u, bg, int(mb["x_length"][i]))
        assert list(u2) == uni and list(b2) == big
