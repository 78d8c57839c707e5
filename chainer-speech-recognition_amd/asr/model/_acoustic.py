"""Shared tail of the AcousticModel classes: turn the stack's (B, V, 1, T) output into what the CTC losses take
(asr/model/cnn.py:33-49, asr/model/sru.py:29-45) and atomic save / load (asr/model/cnn.py:51-63)."""
import os

from .. import nn


def mark_logit_layers(layers):
    """The last normalisation / projection writes float32 logits straight in (T, B, V) order."""
    last = None
    for layer in layers:
        if isinstance(layer, nn.Residual):
            for sub in layer.layers:
                if hasattr(sub, "output_float32"):
                    last = sub
        elif isinstance(layer, nn.GLU):
            last = layer.W if hasattr(layer.W, "output_float32") else last
        elif hasattr(layer, "output_float32"):
            last = layer
    if last is not None:
        last.output_float32 = True
        # a LayerNormalization reads the projection in front of it in float32 as well
        prev = None
        for layer in layers:
            if layer is last and isinstance(last, nn.LayerNormalization) and prev is not None and hasattr(prev, "output_float32"):
                prev.output_float32 = True
            prev = layer


class TimeSteps(tuple):
    """the reference's tuple of T (B, V) Variables, as views of ONE (T, B, V) buffer that the losses pick up without a
    copy (``.buffer``); indexing, len() and iteration behave like the plain tuple"""

    def __new__(cls, tbv):
        self = super(TimeSteps, cls).__new__(cls, tbv.unbind(0))
        self.buffer = tbv
        return self


def split_output(out_data, batchsize, seq_length, split_into_variables):
    """out_data logical (B, V, 1, T) [or (B, V, T)] backed by a (T, B, 1, V) / (T, B, V) buffer."""
    if out_data.dim() == 4:
        assert out_data.shape[3] == seq_length
        tbv = out_data.permute(3, 0, 2, 1).squeeze(2)        # (T, B, V) view of the physical buffer
    else:
        assert out_data.shape[2] == seq_length
        tbv = out_data.permute(2, 0, 1)
    if split_into_variables:
        return TimeSteps(tbv)                                 # T views (B, V): swapaxes/reshape/split_axis of the reference
    return tbv.permute(1, 0, 2)                               # (B, T, V)


def save_atomic(module, filename):
    """asr/model/cnn.py:51-56: written under a temporary name, then renamed; the reference's parameter paths
    (asr/serializers.py)"""
    from .. import serializers
    serializers.save(filename, module)


def load_if_exists(module, filename):
    """asr/model/cnn.py:58-63; works on a freshly built model (lazily sized parameters take the file's shapes)"""
    if os.path.isfile(filename):
        print("Loading {} ...".format(filename))
        from .. import serializers
        serializers.load(filename, module)
        return True
    return False


class AcousticCall(object):
    """mix-in giving a layer container (nn.Stream or nn.Module) the acoustic-model call of the reference
    (asr/model/cnn.py:33-49, asr/model/sru.py:29-45): run the stack on x (B, C, H, T), check that the time axis survived,
    hand the logits over per time step (tuple of T (B, V) views) or as one (B, T, V) view; plus save / load."""

    def __call__(self, x, split_into_variables=True):
        n_utt, n_frames = x.shape[0], x.shape[3]
        mark_logit_layers(self.layers)
        logits = super(AcousticCall, self).__call__(x)
        if logits.shape[3] != n_frames:
            raise AssertionError("the model changed the number of frames: %d -> %d" % (n_frames, logits.shape[3]))
        return split_output(logits, n_utt, n_frames, split_into_variables)

    def save(self, filename):
        save_atomic(self, filename)

    def load(self, filename):
        return load_if_exists(self, filename)
