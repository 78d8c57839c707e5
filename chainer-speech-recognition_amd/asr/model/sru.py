"""Conv + recurrent acoustic model container built from named sub-modules (reference: asr/model/sru.py:9-59)."""
from .. import nn
from .base import NeedsVocabulary
from ._acoustic import AcousticCall


class Configuration(NeedsVocabulary):
    FIELDS = dict(ndim_audio_features=40, ndim_conv=64, ndim_h=128, ndim_dense=256, num_rnn_layers=2, kernel_size=(3, 5),
                  dropout=0)


def configure():
    return Configuration()


class AcousticModel(AcousticCall, nn.Module):
    pass
