"""Operator API of the reference (asr/nn/nn.py) on the HIP path: same names, arguments and call semantics.

Every class cites the reference line it mirrors.  Layers hold float32 master parameters (``chainer``-style
``.W``/``.b``/``.gamma``/``.beta`` attributes) and hand the arithmetic to ``asr.functions``.
"""
import math
import os
import uuid

import torch

from .. import functions
from ..link import Chain, Link, Parameter, get_initializer, initializers  # noqa: F401
from .convolution_1d import Convolution1D
from .convolution_2d import Convolution2D as WeightnormConvolution2D
from .convolution_2d import PlainConvolution2D
from .gru import GRU, NStepBiGRU, NStepGRU, BiGRU  # noqa: F401
from .layernorm import normalize_layer  # noqa: F401
from .sru import SRU  # noqa: F401


# Standard functions (asr/nn/nn.py:11-73)

class ClippedReLU():
    def __init__(self, z=20):
        self.z = z

    def __call__(self, x):
        return functions.clipped_relu(x, self.z)


class ELU():
    def __init__(self, alpha=1):
        self.alpha = alpha

    def __call__(self, x):
        return functions.elu(x, self.alpha)


def HardSigmoid():
    return functions.hard_sigmoid


class LeakyReLU():
    def __init__(self, slope=1):
        self.slope = slope

    def __call__(self, x):
        return functions.leaky_relu(x, self.slope)


class Maxout():
    """asr/nn/nn.py:45-50 (the reference's default pool_size=0.5 is unusable; every call site passes 2)."""

    def __init__(self, pool_size=0.5):
        self.pool_size = pool_size

    def __call__(self, x):
        return functions.maxout(x, self.pool_size)


def ReLU():
    return functions.relu


def Sigmoid():
    return functions.sigmoid


class Softplus():
    def __init__(self, beta=1):
        self.beta = beta

    def __call__(self, x):
        return functions.softplus(x, self.beta)


def Tanh():
    return functions.tanh


# Pooling (asr/nn/nn.py:95-103)

class MaxPooling2D():
    def __init__(self, ksize, stride=None, pad=0, cover_all=True):
        self.ksize = ksize
        self.stride = stride
        self.pad = pad
        self.cover_all = cover_all

    def __call__(self, x):
        # the reference forwards only ksize/stride/pad, so Chainer's default cover_all=True applies (asr/nn/nn.py:102-103)
        return functions.max_pooling_2d(x, self.ksize, self.stride, self.pad)


# Array manipulations (asr/nn/nn.py:148-207) -- views

class BroadcastTo():
    def __init__(self, shape):
        self.shape = shape

    def __call__(self, x):
        return functions.broadcast_to(x, self.shape)


class ExpandDims():
    def __init__(self, axis):
        self.axis = axis

    def __call__(self, x):
        return functions.expand_dims(x, self.axis)


def Flatten():
    return functions.flatten


class Reshape():
    def __init__(self, shape):
        self.shape = shape

    def __call__(self, x):
        return functions.reshape(x, self.shape)


class RollAxis():
    def __init__(self, axis, start=0):
        self.axis = axis
        self.start = start

    def __call__(self, x):
        return functions.rollaxis(x, self.axis, self.start)


class Squeeze():
    def __init__(self, axis):
        self.axis = axis

    def __call__(self, x):
        return functions.squeeze(x, self.axis)


class SwapAxes():
    def __init__(self, axis1, axis2):
        self.axis1 = axis1
        self.axis2 = axis2

    def __call__(self, x):
        return functions.swapaxes(x, self.axis1, self.axis2)


class Tile():
    def __init__(self, reps):
        self.reps = reps

    def __call__(self, x):
        return functions.tile(x, self.reps)


class Transpose():
    def __init__(self, axes):
        self.axes = axes

    def __call__(self, x):
        return functions.transpose(x, self.axes)


# Noise injection (asr/nn/nn.py:211-218)

class Dropout():
    def __init__(self, ratio=0.5):
        self.ratio = ratio

    def __call__(self, x):
        if self.ratio == 0:
            return x
        return functions.dropout(x, self.ratio)


# Links

def Convolution2D(in_channel, out_channel, ksize, stride=1, pad=0, initialW=None, weightnorm=False, causal=False):
    """asr/nn/nn.py:235-238: the stride argument is ignored (always 1), as in the reference.
    ``causal=True`` (extension) makes a layer built with pad=(ph, kw-1) return only the first T time steps, i.e.
    the result of the reference's ``lambda x: x[..., :-pad]`` that follows every such layer, without computing
    the discarded tail."""
    if weightnorm:
        return WeightnormConvolution2D(in_channel, out_channel, ksize, stride=1, pad=pad, initialV=initialW, causal=causal)
    return PlainConvolution2D(in_channel, out_channel, ksize, stride=1, pad=pad, initialW=initialW, causal=causal)


class Linear(Link):
    """chainer.links.Linear (reachable through ``from chainer.links import *``, asr/nn/nn.py:3)."""

    def __init__(self, in_size, out_size=None, nobias=False, initialW=None, initial_bias=None):
        super().__init__()
        if out_size is None:
            in_size, out_size = None, in_size
        self.out_size = out_size
        self._initialW = initialW
        self.W = Parameter()
        self.b = None if nobias else Parameter(get_initializer(0 if initial_bias is None else initial_bias)((out_size,)))
        if in_size is not None:
            self._initialize_params(in_size)

    def _initialize_params(self, in_size):
        self.W.data = get_initializer(self._initialW)((self.out_size, in_size)).to(self.W.device)

    def __call__(self, x):
        if self.W.numel() == 0:
            self._initialize_params(x.shape[1])
        return functions.linear(x, self.W, self.b, self)


class LayerNormalization(Link):
    """asr/nn/nn.py:240-265.  ``eps`` is accepted and ignored, as in the reference (asr/nn/layernorm.py:30-48)."""

    def __init__(self, size=None, eps=1e-6, initial_gamma=None, initial_beta=None):
        super().__init__()
        self._initial_gamma = 1 if initial_gamma is None else initial_gamma
        self._initial_beta = 0 if initial_beta is None else initial_beta
        self.gamma = Parameter()
        self.beta = Parameter()
        self.eps = eps
        self.output_float32 = False       # set by AcousticModel on the layer that feeds the CTC loss
        if size is not None:
            self._initialize_params(size)

    def _initialize_params(self, size):
        self.gamma.data = get_initializer(self._initial_gamma)((size,)).to(self.gamma.device)
        self.beta.data = get_initializer(self._initial_beta)((size,)).to(self.beta.device)

    def __call__(self, x):
        if self.gamma.numel() == 0:
            self._initialize_params(x.shape[1])
        return functions.layer_normalization(x, self.gamma, self.beta, self.output_float32)


class BatchNormalization(Link):
    """chainer.links.BatchNormalization (visible as asr.nn.BatchNormalization through `from chainer.links import *`,
    asr/nn/nn.py:3): per-channel (axis 1) statistics over every other axis, eps 2e-5, running averages with decay 0.9
    (unbiased variance), used instead of the batch statistics when functions.train_mode is off."""

    def __init__(self, size, decay=0.9, eps=2e-5, initial_gamma=None, initial_beta=None):
        super().__init__()
        self.gamma = Parameter(get_initializer(1 if initial_gamma is None else initial_gamma)((size,)))
        self.beta = Parameter(get_initializer(0 if initial_beta is None else initial_beta)((size,)))
        self.register_buffer("avg_mean", torch.zeros(size, dtype=torch.float32))
        self.register_buffer("avg_var", torch.ones(size, dtype=torch.float32))
        self.decay, self.eps = decay, eps

    def __call__(self, x):
        return functions.batch_normalization(x, self.gamma, self.beta, self.avg_mean, self.avg_var, self.eps, self.decay)


class GLU(object):
    """asr/nn/nn.py:267-281."""

    def __init__(self, in_channels, out_channels, ksize=(3, 5), pad=0, wgain=1., weightnorm=False):
        wstd = math.sqrt(wgain / in_channels / ksize[0] / ksize[1])
        self.W = Convolution2D(in_channels, 2 * out_channels, ksize, stride=1, pad=pad,
                               initialW=initializers.HeNormal(wstd), weightnorm=weightnorm)
        self._in_channels, self._out_channels, self._kernel_size, = in_channels, out_channels, ksize

    def __call__(self, X):
        pad = self._kernel_size[1] - 1
        WX = self.W(X)
        if pad > 0:
            WX = WX[..., :-pad]
        return functions.glu(WX)


# Connections (asr/nn/nn.py:285-292)

class Residual(object):
    def __init__(self, *layers):
        self.layers = layers

    def __call__(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


def _apply_layers(layers, x):
    """asr/nn/nn.py:322-328 / 408-414: sequential application; a Residual layer adds its input."""
    for layer in layers:
        y = layer(x)
        if isinstance(layer, Residual):
            y = functions.add(y, x)
        x = y
    return x


# Chains (asr/nn/nn.py:296-414)

class Stream(Chain):
    def __init__(self, *layers):
        super(Stream, self).__init__()
        assert not hasattr(self, "layers")
        self.layers = []
        if len(layers) > 0:
            self.layer(*layers)

    def layer(self, *layers):
        with self.init_scope():
            for i, layer in enumerate(layers):
                index = i + len(self.layers)
                if isinstance(layer, Link):
                    setattr(self, "layer_%d" % index, layer)
                if isinstance(layer, GLU):
                    setattr(self, "layer_%d" % index, layer.W)
                if isinstance(layer, Residual):
                    for _index, _layer in enumerate(layer.layers):
                        if isinstance(_layer, Link):
                            setattr(self, "layer_{}_{}".format(index, _index), _layer)
        self.layers += layers

    def __call__(self, x):
        return _apply_layers(self.layers, x)


class Module(Chain):
    def __init__(self, *layers):
        super(Module, self).__init__()
        self.layers = []
        self.blocks = []
        self._links = []
        self._submodules = []
        self._locked = False
        if len(layers) > 0:
            self.add(*layers)

    def add(self, *layers):
        with self.init_scope():
            for i, layer in enumerate(layers):
                index = i + len(self.layers)
                if isinstance(layer, Link):
                    setattr(self, "_sequential_%d" % index, layer)
                if isinstance(layer, Residual):
                    for _index, _layer in enumerate(layer.layers):
                        if isinstance(_layer, Link):
                            setattr(self, "_sequential_{}_{}".format(index, _index), _layer)
        self.layers += layers
        self.blocks.append(layers)

    def __setattr__(self, name, value):
        if isinstance(value, Module):
            self._submodules.append((name, value))
            value._locked = True
            return super(Module, self).__setattr__(name, value)
        if isinstance(value, Link):
            assert self._locked is False, "Since this module is owned by another module, it is not possible to add Link."
            if not name.startswith("_sequential_"):
                self._links.append((name, value))
        super(Module, self).__setattr__(name, value)

    def save(self, filename):
        tmp_filename = filename + "." + str(uuid.uuid4())
        torch.save(self.state_dict(), tmp_filename)
        if os.path.isfile(filename):
            os.remove(filename)
        os.rename(tmp_filename, filename)

    def load(self, filename):
        if os.path.isfile(filename):
            print("Loading {} ...".format(filename))
            self.load_state_dict(torch.load(filename, map_location="cpu"))
            return True
        return False

    def __call__(self, x):
        return _apply_layers(self.layers, x)
