"""Fully convolutional acoustic model (Zhang et al., https://arxiv.org/abs/1701.02720) -- the container the recipes of
run/ctc/cnn/model.py fill with layers (reference: asr/model/cnn.py:9-63)."""
from .. import nn
from .base import NeedsVocabulary
from ._acoustic import AcousticCall


class Configuration(NeedsVocabulary):
    FIELDS = dict(ndim_audio_features=40, ndim_h=128, ndim_dense=256, num_conv_layers=5, kernel_size=(3, 5), dropout=0,
                  weightnorm=False, wgain=1, architecture="zhang")


def configure():
    return Configuration()


class AcousticModel(AcousticCall, nn.Stream):
    pass
