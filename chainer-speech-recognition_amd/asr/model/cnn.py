"""CNN acoustic model -- asr/model/cnn.py:9-63 (Configuration, AcousticModel(nn.Stream))."""
from .. import nn
from . import base
from ._acoustic import load_if_exists, mark_logit_layers, save_atomic, split_output


class Configuration(base.Configuration):
    def __init__(self):
        super().__init__()
        self.vocab_size = -1
        self.ndim_audio_features = 40
        self.ndim_h = 128
        self.ndim_dense = 256
        self.num_conv_layers = 5
        self.kernel_size = (3, 5)
        self.dropout = 0
        self.weightnorm = False
        self.wgain = 1
        self.architecture = "zhang"

    def save(self, filename):
        assert self.vocab_size > 0
        super().save(filename)


def configure():
    return Configuration()


# Towards End-to-End Speech Recognition with Deep Convolutional Neural Networks  https://arxiv.org/abs/1701.02720
class AcousticModel(nn.Stream):
    def __call__(self, x, split_into_variables=True):
        batchsize = x.shape[0]
        seq_length = x.shape[3]
        mark_logit_layers(self.layers)
        out_data = super(AcousticModel, self).__call__(x)
        assert out_data.shape[3] == seq_length
        return split_output(out_data, batchsize, seq_length, split_into_variables)

    def save(self, filename):
        save_atomic(self, filename)

    def load(self, filename):
        return load_if_exists(self, filename)
