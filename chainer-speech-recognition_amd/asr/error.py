"""Greedy-decode error rates with the reference's function names (asr/error.py:7-68).

``compute_minibatch_error`` keeps the reference's signature.  Given device tensors it runs the blank / repeat collapse and
the Levenshtein distances of the whole minibatch on the GPU (``libasr_hip``: asr_ctc_collapse, asr_edit_distance) and brings
back only one distance and two lengths per utterance; given host arrays it moves them to the GPU first.  There is no CPU
implementation of the batch path here (the CPU restatement lives in ``oracle/text.py``, test infrastructure).
"""
import numpy as np
import torch

from . import _ops
from .vocab import convert_sentence_to_unigram_ids


def compute_character_error_rate(r, h):
    """Levenshtein(r, h) / len(r) for one pair of id sequences; len(h) when r is empty (asr/error.py:7-24).
    Goes through the same device kernel as the batch path.  (The reference keeps its table in uint8, i.e. it is defined
    for sequences of up to 255 tokens; this one is exact beyond.)"""
    if len(r) == 0:
        return len(h)
    dev = torch.device("cuda", torch.cuda.current_device())
    rr = torch.tensor([list(r)], dtype=torch.int32, device=dev)
    hh = torch.tensor([list(h) if len(h) else [0]], dtype=torch.int32, device=dev)
    d = _ops.edit_distance(rr, torch.tensor([len(r)], dtype=torch.int32, device=dev), hh,
                           torch.tensor([len(h)], dtype=torch.int32, device=dev))
    return float(d.item()) / len(r)


def greedy_decode(logits, blank=0, lengths=None):
    """(T, B, V) f32 logits on the GPU -> (ids (B, T) int32 padded with blank, lengths (B) int32): argmax per frame,
    repeats merged, blanks dropped (run/ctc/cnn/dev.py:106 + asr/error.py:38-47).  `lengths` (frames per utterance)
    restricts the decode to the valid frames; the reference decodes all T frames (lengths=None)."""
    ids = _ops.argmax_rows(logits.contiguous())
    return _ops.ctc_collapse(ids, lengths, blank, True)


def _needs_retokenisation(vocab_id_to_token, vocab_token_to_id):
    """The reference turns the predicted ids into a sentence and tokenises it again (asr/error.py:49-53): an identity
    unless some token of the inventory is spelled by several unigrams (bigram entries of the Gram-CTC inventory)."""
    if vocab_id_to_token is None or vocab_token_to_id is None:
        return False
    for tid, tok in vocab_id_to_token.items():
        if tid == 0 or tok == "_":
            continue
        if convert_sentence_to_unigram_ids(tok, vocab_token_to_id) != [tid]:
            return True
    return False


def compute_minibatch_error(y_batch, t_batch, BLANK, vocab_token_to_id, vocab_id_to_token, print_sequences=False):
    """y_batch (B, T): argmax ids per frame; t_batch (B, L): labels padded with BLANK.  Mean over the minibatch of
    Levenshtein(pred, true) / len(true)  (len(pred) where the transcription is empty)."""
    dev = torch.device("cuda", torch.cuda.current_device())
    y = torch.as_tensor(np.asarray(y_batch) if not isinstance(y_batch, torch.Tensor) else y_batch).to(dev, torch.int32).contiguous()
    t = torch.as_tensor(np.asarray(t_batch) if not isinstance(t_batch, torch.Tensor) else t_batch).to(dev, torch.int32).contiguous()
    pred, pred_len = _ops.ctc_collapse(y, None, BLANK, True)
    true, true_len = _ops.ctc_collapse(t, None, BLANK, False)
    if _needs_retokenisation(vocab_id_to_token, vocab_token_to_id):
        # string work: inherently host side (only for inventories with multi-unigram tokens)
        ph, pl = pred.cpu().numpy(), pred_len.cpu().numpy()
        rows = []
        for b in range(ph.shape[0]):
            sentence = "".join(vocab_id_to_token[int(i)] for i in ph[b, :pl[b]])
            rows.append(convert_sentence_to_unigram_ids(sentence, vocab_token_to_id))
        width = max(1, max(len(r) for r in rows))
        host = np.full((len(rows), width), BLANK, dtype=np.int32)
        for b, r in enumerate(rows):
            host[b, :len(r)] = r
        pred = torch.from_numpy(host).to(dev)
        pred_len = torch.tensor([len(r) for r in rows], dtype=torch.int32, device=dev)
    dist = _ops.edit_distance(true, true_len, pred, pred_len)
    d, n = dist.cpu().numpy().astype(np.float64), true_len.cpu().numpy()
    _ops.gru_check_all()        # the ids came from a forward pass nobody else checks: raise if a recurrence gave up a wait
    per = np.where(n > 0, d / np.maximum(n, 1), d)         # len(r) == 0: the distance is len(h), returned as it is
    if print_sequences and vocab_id_to_token is not None:
        ph, pl, th, tl = pred.cpu().numpy(), pred_len.cpu().numpy(), true.cpu().numpy(), true_len.cpu().numpy()
        for b in range(ph.shape[0]):
            print("#{}".format(b + 1))
            print("pred:\t" + "".join(vocab_id_to_token[int(i)] for i in ph[b, :pl[b]]))
            print("true:\t" + "".join(vocab_id_to_token[int(i)] for i in th[b, :tl[b]]))
    return float(per.sum() / len(per))
