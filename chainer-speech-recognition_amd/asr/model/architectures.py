"""The seven CNN recipes of run/ctc/cnn/model.py:11-332 (``build_model(config)``), written against this package's nn.

Same layer order, channel counts, paddings, initialisers and parameter names (``layer_%d`` / ``layer_%d_%d``) as the
reference; the only difference is that a time-padded convolution is built with ``causal=True`` and the slice that
follows it in the reference (``lambda x: x[..., :-pad]``, run/ctc/cnn/model.py:44) becomes an identity kept at the same
position in the layer list, so the discarded tail is never computed and the layer indices -- hence the checkpoint
names -- stay put.
"""
import math

from .. import nn
from ..link import initializers
from .cnn import AcousticModel

ARCHITECTURES = ("zhang", "zhang+fc_relu", "zhang+residual", "zhang+layernorm", "glu", "relu+layernorm",
                 "relu+layernorm+residual")


def _crop(x):
    """position of the reference's ``x[..., :-pad]``; the causal convolution in front of it already returned T steps"""
    return x


_crop._asr_identity = True      # (asr.nn containers look through it when they fuse a convolution with the layers behind it)


def build_model(config):
    V, cin, h, dense = config.vocab_size, config.ndim_audio_features, config.ndim_h, config.ndim_dense
    ks = tuple(config.kernel_size)
    nconv, drop, wn, wgain, arch = config.num_conv_layers, config.dropout, config.weightnorm, config.wgain, config.architecture
    for name, typ in (("vocab_size", int), ("ndim_audio_features", int), ("ndim_h", int), ("ndim_dense", int),
                      ("num_conv_layers", int), ("weightnorm", bool), ("architecture", str)):
        assert isinstance(getattr(config, name), typ), name
    if arch not in ARCHITECTURES:
        raise NotImplementedError()
    pad_t = ks[1] - 1
    kernel_height = int(math.ceil((config.num_mel_filters - 2) / 3))          # run/ctc/cnn/model.py:38

    def normal(fan_c):          # initializers.Normal(sqrt(wgain / C / kh / kw))  (:145, :162, ...)
        return initializers.Normal(math.sqrt(wgain / fan_c / ks[0] / ks[1]))

    def conv(ci, co, pad_h, init=None):
        return nn.Convolution2D(ci, co, ks, stride=1, pad=(pad_h, pad_t), initialW=init, weightnorm=wn, causal=True)

    def point(ci, co, ksize):   # the "dense" convolutions: no padding, default initialiser
        return nn.Convolution2D(ci, co, ksize=ksize, stride=1, pad=0, weightnorm=wn)

    model = AcousticModel()
    maxout_family = arch in ("zhang", "zhang+fc_relu", "zhang+residual", "zhang+layernorm", "glu")
    explicit_init = arch not in ("zhang", "zhang+fc_relu")
    ln = arch in ("zhang+layernorm", "relu+layernorm", "relu+layernorm+residual")
    width = 2 if maxout_family else 1                                          # maxout halves the channels again

    def act():
        return nn.Maxout(2) if maxout_family else nn.ReLU()

    # first layer (:42-48 and its siblings)
    first = [conv(cin, h * width, 0, normal(cin) if explicit_init else None), _crop]
    if ln:
        first.append(nn.LayerNormalization(None))
    first += [act(), nn.Dropout(drop), nn.MaxPooling2D(ksize=(3, 1))]
    model.layer(*first)

    dense_in = h
    if arch in ("zhang", "zhang+fc_relu", "zhang+residual"):
        narrow = min(nconv, 4)
        wide = max(0, nconv - 4)
        in_out = [(h, h * 2)] * narrow
        if wide > 0:
            in_out[-1] = (h, h * 4)                                            # :53-54
        residual = arch == "zhang+residual"
        for idx, (ci, co) in enumerate(in_out):
            block = [conv(ci, co, 1, normal(h) if explicit_init else None), _crop, nn.Maxout(2), nn.Dropout(drop)]
            if residual and idx != len(in_out) - 1:                            # :160-176: the last narrow layer is plain
                model.layer(nn.Residual(*block))
            else:
                model.layer(*block)
        if wide > 0:
            in_out = [(h * 2, h * 4)] * narrow                                 # :64-72 / :178-187 (loop count = narrow)
            for ci, co in in_out:
                block = [conv(ci, co, 1, normal(h) if explicit_init else None), _crop, nn.Maxout(2), nn.Dropout(drop)]
                model.layer(nn.Residual(*block)) if residual else model.layer(*block)
        dense_in = in_out[-1][0]
    elif arch == "glu":
        for _ in range(nconv):
            model.layer(nn.GLU(h, h, ks, pad=(1, pad_t), weightnorm=wn), nn.Dropout(drop))
    elif arch == "relu+layernorm+residual":
        for _ in range(nconv):                                                 # :315-323: pre-activation residual block
            model.layer(nn.Residual(nn.LayerNormalization(None), nn.ReLU(), nn.Dropout(drop), conv(h, h, 1, normal(h)), _crop))
    else:                                                                      # zhang+layernorm, relu+layernorm
        for _ in range(nconv):
            model.layer(conv(h, h * width, 1, normal(h)), _crop, nn.LayerNormalization(None), act(), nn.Dropout(drop))

    # dense layers
    if arch == "glu":
        model.layer(nn.GLU(h, dense, ksize=(kernel_height, 1), pad=0, weightnorm=wn), nn.Dropout(drop))
    elif arch == "zhang+fc_relu":
        model.layer(point(dense_in, dense, (kernel_height, 1)), nn.ReLU(), nn.Dropout(drop))
        model.layer(point(dense, dense, 1), nn.ReLU(), nn.Dropout(drop))
    elif arch in ("zhang", "zhang+residual"):
        model.layer(point(dense_in, dense * 2, (kernel_height, 1)), nn.Maxout(2), nn.Dropout(drop))
        model.layer(point(dense, dense * 2, 1), nn.Maxout(2), nn.Dropout(drop))
    else:
        model.layer(point(h, dense * width, (kernel_height, 1)), nn.LayerNormalization(None), act(), nn.Dropout(drop))
    model.layer(point(dense, V, 1), nn.LayerNormalization(None))
    return model
