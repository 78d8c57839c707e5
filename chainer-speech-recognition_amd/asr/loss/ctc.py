"""CTC and Gram-CTC losses on the HIP path.

Drop-in for (reference file:line)
  * ``chainer.functions.connectionist_temporal_classification(x, t, blank_symbol, input_length,
    label_length, reduce)`` -- call sites run/ctc/cnn/train.py:162,191
  * ``asr.loss.gram_ctc(xs, label_unigram, label_bigram, blank_symbol, input_length, length_unigram,
    reduce)`` -- asr/loss/gram_ctc.py:300-315

``xs`` may be the reference's tuple/list of T arrays (B, V) or ONE (T, B, V) tensor (what the models
of this package return with ``split_into_variables=True``: a tuple of views of one buffer, which is
recognised and used without a copy).
"""
import torch

from .. import _lib


def _as_tbv(xs):
    """tuple of T (B, V) views of one contiguous buffer -> that (T, B, V) buffer, else stack."""
    if isinstance(xs, torch.Tensor):
        if xs.dim() != 3:
            raise TypeError("xs must be (T, B, V)")
        return xs
    if not isinstance(xs, (list, tuple)) or len(xs) == 0:
        raise TypeError("xs must be a list of Variables")       # asr/loss/gram_ctc.py:301-302
    whole = getattr(xs, "buffer", None)         # asr.model TimeSteps: the views' (T, B, V) buffer
    if whole is not None and whole.dim() == 3 and whole.shape[0] == len(xs):
        return whole
    base = getattr(xs[0], "_base", None)
    if base is not None and base.is_contiguous() and base.numel() == len(xs) * xs[0].numel() and xs[0].is_contiguous() \
            and xs[0].data_ptr() == base.data_ptr() \
            and xs[-1].data_ptr() == base.data_ptr() + (len(xs) - 1) * xs[0].numel() * base.element_size():
        return base.view((len(xs),) + tuple(xs[0].shape))
    return torch.stack(list(xs), dim=0)


class _CTCFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xs, label_unigram, label_bigram, input_length, label_length, blank, reduce, box):
        lib = _lib.lib()
        if xs.dtype != torch.float32:
            raise TypeError("xs must be float32")                 # asr/loss/gram_ctc.py:241-244
        for t in (label_unigram, label_bigram, input_length, label_length):
            if t is not None and t.dtype != torch.int32:
                raise TypeError("labels and lengths must be int32")   # asr/loss/gram_ctc.py:234-235
        xs = xs.contiguous()
        _lib.ptr(xs)                # raises on a CPU tensor: there is no CPU path
        T, B, V = xs.shape
        Lmax = label_unigram.shape[1]
        gram = label_bigram is not None
        if gram and label_bigram.shape != label_unigram.shape:
            raise ValueError("label_unigram and label_bigram must have the same shape")   # asr/loss/gram_ctc.py:308
        label_unigram = label_unigram.contiguous()
        label_bigram = label_bigram.contiguous() if gram else None
        nbytes = lib.asr_ctc_workspace_bytes(T, B, V, Lmax, int(gram))
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xs.device)
        loss_b = torch.empty(B, dtype=torch.float32, device=xs.device)
        loss_m = torch.empty((), dtype=torch.float32, device=xs.device)
        # logits straight out of a per-frame LayerNormalization come with the log-sum-exp of every row (functions._CtcBox.lse)
        row_lse = box.lse if (box is not None and box.lse is not None and box.lse.numel() == T * B) else None
        rc = lib.asr_ctc_forward_lse(_lib.stream(), _lib.ptr(xs), _lib.ptr(label_unigram), _lib.ptr(label_bigram),
                                     _lib.ptr(input_length), _lib.ptr(label_length), T, B, V, Lmax, int(blank),
                                     _lib.ptr(loss_b), _lib.ptr(loss_m), _lib.ptr(ws), nbytes, _lib.ptr(row_lse))
        _lib.check(rc, "asr_ctc_forward_lse")
        ctx.save_for_backward(xs, input_length, ws)
        ctx.dims = (T, B, V, Lmax, int(gram), reduce)
        ctx.box = box if Lmax * (3 if gram else 2) + 1 <= 512 else None
        return loss_m if reduce == "mean" else loss_b

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.lib()
        xs, input_length, ws = ctx.saved_tensors
        T, B, V, Lmax, gram, reduce = ctx.dims
        gy = gy.contiguous().to(torch.float32)
        scale = 1.0 / B if reduce == "mean" else 1.0
        if ctx.box is not None:
            # the logits come straight out of a per-frame LayerNormalization: leave the recipe there; its backward forms
            # (softmax - occupancy) * scale * gy in registers instead of reading a (T, B, V) float32 gradient (functions._CtcBox)
            from ..functions import _zero_token
            first = ctx.box.post(dict(ws=ws, Lmax=Lmax, gram=gram, x_len=input_length, gy=gy, gy_per_utt=0 if reduce == "mean" else 1,
                                      scale=scale))
            return (_zero_token(xs.shape, xs.device) if first else None), None, None, None, None, None, None, None
        grad = torch.empty_like(xs)
        rc = lib.asr_ctc_backward(_lib.stream(), _lib.ptr(xs), _lib.ptr(input_length), T, B, V, Lmax, gram,
                                  _lib.ptr(gy), 0 if reduce == "mean" else 1, scale, _lib.ptr(grad), _lib.ptr(ws),
                                  ws.numel())
        _lib.check(rc, "asr_ctc_backward")
        return grad, None, None, None, None, None, None, None


def _check_common(xs, blank_symbol, reduce):
    if not isinstance(blank_symbol, int):
        raise TypeError("blank_symbol must be non-negative integer.")     # asr/loss/gram_ctc.py:303-304
    if reduce not in ("mean", "no"):
        raise ValueError("only 'mean' and 'no' are valid for 'reduce', but '%s' is given" % reduce)  # :224-227
    x = _as_tbv(xs)
    assert blank_symbol >= 0
    assert blank_symbol < x.shape[2]
    return x


def connectionist_temporal_classification(x, t, blank_symbol, input_length=None, label_length=None, reduce="mean"):
    """CTC loss with Chainer's conventions: mean over the batch of -log p (not divided by T)."""
    xs = _check_common(x, blank_symbol, reduce)
    from ..functions import ctc_box_of
    return _CTCFunction.apply(xs, t, None, input_length, label_length, blank_symbol, reduce, ctc_box_of(xs))


def gram_ctc(xs, label_unigram, label_bigram, blank_symbol, input_length=None, length_unigram=None, reduce="mean"):
    """Gram-CTC loss over the unigram + bigram lattice (asr/loss/gram_ctc.py:300-315)."""
    x = _check_common(xs, blank_symbol, reduce)
    assert label_unigram.shape[1] == label_bigram.shape[1]
    from ..functions import ctc_box_of
    return _CTCFunction.apply(x, label_unigram, label_bigram, input_length, length_unigram, blank_symbol, reduce, ctc_box_of(x))
