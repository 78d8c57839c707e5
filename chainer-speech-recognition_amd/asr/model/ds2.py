"""Deep-Speech-2-style conv + (Bi)GRU CTC acoustic model -- the model BASELINE.json's configs[0..2] name.

The reference ships no GRU model; this one is written against the `asr.nn` API in the shape of the reference's
conv + recurrent template run/ctc/sru/model.py:10-139 (conv blocks -> reshape (B, C*H, T) -> recurrent blocks ->
1x1 dense blocks with Maxout -> LayerNormalization -> per-time-step split), with the recurrent output actually fed
forward (the template drops it, run/ctc/sru/model.py:120-124).  SURVEY.md section 8(d) fixes the sizes:
  conv1 3->2*ndim_conv k(3,5) -> Maxout -> MaxPool(3,1);  conv2 ndim_conv->2*ndim_conv -> Maxout -> MaxPool(2,1);
  num_rnn_layers x BiGRU(ndim_rnn) (directions summed);  Conv1D(->2*dense)+Maxout, Conv1D(dense->2*dense)+Maxout,
  Conv1D(dense->V), LayerNormalization.
"""
from .. import functions, nn
from .base import NeedsVocabulary
from ._acoustic import load_if_exists, save_atomic, split_output


class Configuration(NeedsVocabulary):
    FIELDS = dict(ndim_audio_features=3, ndim_conv=64, ndim_rnn=512, ndim_dense=320, num_conv_layers=2, num_rnn_layers=4,
                  bidirectional=True, kernel_size=(3, 5), dropout=0)


def configure():
    return Configuration()


class Model(nn.Module):
    def __init__(self, config):
        super(Model, self).__init__()
        kernel_size = tuple(config.kernel_size)
        pad = kernel_size[1] - 1
        dropout = config.dropout
        self.num_rnn_layers = config.num_rnn_layers

        conv_blocks = nn.Module()
        pools = [3] + [2] * (config.num_conv_layers - 1)
        in_ch = config.ndim_audio_features
        for pool in pools:
            conv_blocks.add(
                # causal=True == pad=(0, kw-1) followed by the reference's `lambda x: x[..., :-pad]`
                nn.Convolution2D(in_ch, config.ndim_conv * 2, kernel_size, stride=1, pad=(0, pad), causal=True),
                nn.Maxout(2),
                nn.Dropout(dropout),
                nn.MaxPooling2D(ksize=(pool, 1)),
            )
            in_ch = config.ndim_conv
        self.conv_blocks = conv_blocks
        height = config.num_mel_filters
        for pool in pools:          # pad_h = 0 convolution, then cover_all pooling
            height -= kernel_size[0] - 1
            height = 1 if height <= pool else -(-(height - pool) // pool) + 1
        self._merged = (config.ndim_conv, height)

        rnn_blocks = nn.Module()
        rnn_cls = nn.BiGRU if config.bidirectional else nn.GRU
        for _ in range(config.num_rnn_layers):
            rnn_blocks.add(rnn_cls(None, config.ndim_rnn), nn.Dropout(dropout))
        self.rnn_blocks = rnn_blocks

        dense_blocks = nn.Module()
        dense_blocks.add(nn.Convolution1D(None, config.ndim_dense * 2), nn.Maxout(2), nn.Dropout(dropout))
        dense_blocks.add(nn.Convolution1D(config.ndim_dense, config.ndim_dense * 2), nn.Maxout(2), nn.Dropout(dropout))
        logits = nn.Convolution1D(config.ndim_dense, config.vocab_size)
        norm = nn.LayerNormalization()
        logits.output_float32 = True
        norm.output_float32 = True
        dense_blocks.add(logits, _PerFrame(norm))
        self.dense_blocks = dense_blocks

    def __call__(self, x, split_into_variables=True, x_length=None):
        """x_length (B) int32 on the device (the Loader's x_length_batch, asr/data/loaders/base.py:30): the recurrent layers then
        run every utterance over its own frames (the reverse direction starts at the utterance's last frame, not in the padding);
        None = the padded block as it is."""
        batchsize = x.shape[0]
        seq_length = x.shape[3]
        out_data = self.conv_blocks(x)
        out_data = functions.reshape(out_data, (batchsize, -1, seq_length))
        if x_length is None:
            out_data = self.rnn_blocks(out_data)
        else:
            for layer in self.rnn_blocks.layers:
                out_data = layer(out_data, x_length) if isinstance(layer, (nn.GRU, nn.BiGRU)) else layer(out_data)
        out_data = self.dense_blocks(out_data)
        assert out_data.shape[2] == seq_length
        return split_output(out_data, batchsize, seq_length, split_into_variables)

    def column_permutations(self):
        """parameters whose last axis indexes the merged (channel, height) features of the conv stack: {name: (C, H)}.
        The reference's ``reshape(out, (B, -1, T))`` (run/ctc/sru/model.py:114) orders them (c, h), the physical layout here
        (h, c) -- asr/serializers.py stores the reference's order."""
        return {"rnn_blocks._sequential_0.w_ih": (self._merged[0], self._merged[1])}

    def save(self, filename):
        save_atomic(self, filename)

    def load(self, filename):
        return load_if_exists(self, filename)


class _PerFrame(nn.Link):
    """LayerNormalization over the vocabulary of every frame: views (B, V, T) as the reference's 4-d (B, V, 1, T)
    so that axes (1, 2) are (V, 1) -- the statistics the CNN models use (asr/model/cnn.py + asr/nn/nn.py:260-265) --
    rather than (V, T), which would make an utterance's logits depend on its padding."""

    def __init__(self, norm):
        super().__init__()
        self.norm = norm

    def __call__(self, x):
        return self.norm(x.unsqueeze(2)).squeeze(2)


def build_model(config):
    return Model(config)
