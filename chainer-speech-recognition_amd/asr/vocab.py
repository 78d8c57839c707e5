"""Token inventory and sentence -> token ids, with the reference's function names (asr/vocab.py:63-126).

The kana inventory itself is data: it is read from ``vocab_tables.json`` (exported from a reference checkout by
``tools/export_vocab.py``: token order defines the ids, so a checkpoint trained with the reference keeps its meaning).
The tokeniser is restated here:

  1. a small kana (``SUTEGANA``) joins the character in front of it                       asr/vocab.py:108-115
  2. a joined token found in ``UNIGRAM_COLLAPSE`` is replaced by its plain spelling       :116-118
  3. the replacements are split into characters again, small kana joining as in step 1    :119-126
"""
import json
import os

_TABLES = None


def _tables():
    global _TABLES, UNIGRAM_TOKENS, SUTEGANA, UNIGRAM_COLLAPSE
    if _TABLES is None:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "vocab_tables.json"), encoding="utf-8") as f:
            _TABLES = json.load(f)
    return _TABLES


ID_BLANK = 0
UNIGRAM_TOKENS = list(_tables()["unigram_tokens"])
SUTEGANA = list(_tables()["sutegana"])
UNIGRAM_COLLAPSE = dict(_tables()["collapse"])


def get_unigram_ids():
    """token -> id with "_" = blank = 0, and the inverse map (asr/vocab.py:63-76)."""
    ids = {"_": ID_BLANK}
    for tok in UNIGRAM_TOKENS:
        ids[tok] = len(ids)
    return ids, {v: k for k, v in ids.items()}


def load_unigram_and_bigram_ids(filename):
    """unigram ids followed by one extra id per line of `filename` (the bigram inventory, asr/vocab.py:78-90)."""
    assert os.path.isfile(filename)
    ids, _ = get_unigram_ids()
    with open(filename, "r", encoding="utf-8") as f:
        for line in f:
            ids[line.strip()] = len(ids)
    return ids, {v: k for k, v in ids.items()}


def get_all_bigram_tokens():
    return [(a, b) for a in UNIGRAM_TOKENS for b in UNIGRAM_TOKENS]


def _join_small_kana(chars):
    out = []
    for ch in chars:
        if ch in SUTEGANA:
            assert len(out) > 0, "a sentence cannot start with a small kana"
            out[-1] += ch
        else:
            out.append(ch)
    return out


def convert_sentence_to_unigram_tokens(sentence):
    joined = [UNIGRAM_COLLAPSE.get(tok, tok) for tok in _join_small_kana(sentence)]
    return _join_small_kana(ch for tok in joined for ch in tok)


def convert_sentence_to_unigram_ids(sentence, unigram_token_ids):
    ids = []
    for tok in convert_sentence_to_unigram_tokens(sentence):
        assert tok in unigram_token_ids, tok
        ids.append(unigram_token_ids[tok])
    return ids
