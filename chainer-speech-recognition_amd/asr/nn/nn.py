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


# ---------------------------------------------------------------------------------------------------------------------
# Function layers (asr/nn/nn.py:11-218): stateless callables that remember a few arguments.  They are generated from one
# table -- (layer name, function applied, ((argument, default), ...)) -- instead of one hand-written class each; an
# argument without default is required, arguments may be given by position or keyword, and every argument is kept as an
# attribute of the same name (the recipes read e.g. ``layer.ratio``).
_REQUIRED = object()


def _function_layer(name, fn, signature, doc=None):
    names = tuple(n for n, _ in signature)

    def __init__(self, *args, **kwargs):
        if len(args) > len(names):
            raise TypeError("%s() takes at most %d arguments" % (name, len(names)))
        given = dict(zip(names, args))
        for key, value in kwargs.items():
            if key not in names or key in given:
                raise TypeError("%s() got an unexpected or repeated argument %r" % (name, key))
            given[key] = value
        for key, default in signature:
            if key not in given and default is _REQUIRED:
                raise TypeError("%s() missing argument %r" % (name, key))
            setattr(self, key, given.get(key, default))

    def __call__(self, x):
        return fn(x, *(getattr(self, key) for key in names))

    return type(name, (object,), {"__init__": __init__, "__call__": __call__, "__doc__": doc or "functions.%s" % fn.__name__})


def _plain(fn):
    """layers without arguments are factories returning the function itself (asr/nn/nn.py:25-27,52-56,71-73,159-161)"""
    def factory():
        return fn
    factory.__name__ = fn.__name__
    return factory


def _max_pool(x, ksize, stride, pad, cover_all):
    # the reference forwards only ksize / stride / pad, so Chainer's default cover_all=True always applies (:102-103)
    return functions.max_pooling_2d(x, ksize, stride, pad)


def _dropout(x, ratio):
    return x if ratio == 0 else functions.dropout(x, ratio)


def _gaussian_noise(x, mean, std):
    return functions.gaussian_noise(x, std)        # asr/nn/nn.py:220-231 never uses `mean` either


def _max_pool_nd(x, ksize, stride, pad, cover_all):
    return functions.max_pooling_nd(x, ksize, stride, pad)


for _name, _fn, _sig, _doc in (
        # activations (asr/nn/nn.py:11-73)
        ("ClippedReLU", functions.clipped_relu, (("z", 20),), None),
        ("ELU", functions.elu, (("alpha", 1),), None),
        ("LeakyReLU", functions.leaky_relu, (("slope", 1),), None),
        ("Maxout", functions.maxout, (("pool_size", 0.5),),
         "asr/nn/nn.py:45-50 (the reference's default pool_size=0.5 is unusable; every call site passes 2)"),
        ("Softplus", functions.softplus, (("beta", 1),), None),
        ("CReLU", functions.crelu, (("axis", 1),), "asr/nn/nn.py:18-23"),
        ("Softmax", functions.softmax, (("axis", 1),), "asr/nn/nn.py:58-63"),
        # pooling (:95-103)
        ("MaxPooling2D", _max_pool, (("ksize", _REQUIRED), ("stride", None), ("pad", 0), ("cover_all", True)), "asr/nn/nn.py:95-103"),
        ("MaxPoolingND", _max_pool_nd, (("ksize", _REQUIRED), ("stride", None), ("pad", 0), ("cover_all", True)), "asr/nn/nn.py:105-113"),
        ("AveragePooling2D", functions.average_pooling_2d, (("ksize", _REQUIRED), ("stride", None), ("pad", 0)), "asr/nn/nn.py:77-84"),
        ("AveragePoolingND", functions.average_pooling_nd, (("ksize", _REQUIRED), ("stride", None), ("pad", 0)), "asr/nn/nn.py:86-93"),
        ("SpatialPyramidPooling2D", functions.spatial_pyramid_pooling_2d, (("pyramid_height", _REQUIRED), ("pooling_class", _REQUIRED)),
         "asr/nn/nn.py:115-121 (max pooling, as Chainer)"),
        ("Unpooling2D", functions.unpooling_2d, (("ksize", _REQUIRED), ("stride", None), ("pad", 0), ("outsize", None), ("cover_all", True)),
         "asr/nn/nn.py:123-133"),
        ("UpSampling2D", functions.upsampling_2d, (("indexes", _REQUIRED), ("ksize", _REQUIRED), ("stride", None), ("pad", 0), ("outsize", None),
                                                   ("cover_all", True)), "asr/nn/nn.py:135-146 (indexes: functions.max_pooling_2d_indexes)"),
        # array manipulation (:148-207): views of the physical buffer
        ("BroadcastTo", functions.broadcast_to, (("shape", _REQUIRED),), None),
        ("ExpandDims", functions.expand_dims, (("axis", _REQUIRED),), None),
        ("Reshape", functions.reshape, (("shape", _REQUIRED),), None),
        ("RollAxis", functions.rollaxis, (("axis", _REQUIRED), ("start", 0)), None),
        ("Squeeze", functions.squeeze, (("axis", _REQUIRED),), None),
        ("SwapAxes", functions.swapaxes, (("axis1", _REQUIRED), ("axis2", _REQUIRED)), None),
        ("Tile", functions.tile, (("reps", _REQUIRED),), None),
        ("Transpose", functions.transpose, (("axes", _REQUIRED),), None),
        # noise (:211-218): identity when the ratio is 0
        ("Dropout", _dropout, (("ratio", 0.5),), "asr/nn/nn.py:211-218"),
        ("GaussianNoise", _gaussian_noise, (("mean", _REQUIRED), ("std", _REQUIRED)), "asr/nn/nn.py:220-231")):
    globals()[_name] = _function_layer(_name, _fn, _sig, _doc)

HardSigmoid = _plain(functions.hard_sigmoid)
LogSoftmax = _plain(functions.log_softmax)
ReLU = _plain(functions.relu)
Sigmoid = _plain(functions.sigmoid)
Tanh = _plain(functions.tanh)
Flatten = _plain(functions.flatten)


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


def _fusable_pool(layers, i):
    """index of the MaxPooling2D((k, 1)) that follows the Maxout(2) at layers[i] (identity Dropout(0) layers in between
    are skipped), or -1"""
    if type(layers[i]).__name__ != "Maxout" or layers[i].pool_size != 2:
        return -1
    j = i + 1
    while j < len(layers) and type(layers[j]).__name__ == "Dropout" and layers[j].ratio == 0:
        j += 1
    if j >= len(layers) or type(layers[j]).__name__ != "MaxPooling2D":
        return -1
    pool = layers[j]
    ks = pool.ksize if isinstance(pool.ksize, (tuple, list)) else (pool.ksize, pool.ksize)
    if len(ks) != 2 or ks[1] != 1 or pool.stride is not None or pool.pad not in (0, (0, 0)):
        return -1
    return j


def _apply_layers(layers, x):
    """asr/nn/nn.py:322-328 / 408-414: sequential application; a Residual layer adds its input.  Maxout(2) directly
    followed by MaxPooling2D((k, 1)) runs as one pass (functions.maxout_max_pooling): same values, one read of the
    convolution output instead of a write and two reads more."""
    i = 0
    while i < len(layers):
        layer = layers[i]
        if x.dim() == 4 and hasattr(layer, "fused_maxout_pool"):
            # a model's first convolution directly followed by Maxout(2) and MaxPooling2D((k, 1)): one pass, the convolution's output is
            # never written (functions.convolution_maxout_pool; None where that does not serve the layer)
            m = i + 1
            while m < len(layers) and getattr(layers[m], "_asr_identity", False):
                m += 1
            j = _fusable_pool(layers, m) if m < len(layers) else -1
            if j > 0:
                ks = layers[j].ksize
                y = layer.fused_maxout_pool(x, ks[0] if isinstance(ks, (tuple, list)) else ks)
                if y is not None:
                    x = y
                    i = j + 1
                    continue
        j = _fusable_pool(layers, i) if x.dim() == 4 else -1
        if j > 0:
            ks = layers[j].ksize
            # x was produced by the previous layer of THIS sequence and nothing else holds it: the fused backward may take over
            # the bias gradient of a convolution in front (functions._BiasBox)
            x = functions.maxout_max_pooling(x, ks[0] if isinstance(ks, (tuple, list)) else ks, sole_consumer=i > 0)
            i = j + 1
            continue
        y = layer(x)
        if isinstance(layer, Residual):
            y = functions.add(y, x)
        x = y
        i += 1
    return x


# Chains (asr/nn/nn.py:296-414).  Both containers register the parameterised layers they are given as attributes named
# <prefix><index> (and <prefix><index>_<inner index> for the layers inside a Residual), which is what fixes the parameter
# names of a saved model: "layer_" for Stream (:304-320), "_sequential_" for Module (:341-353).

def _register_layers(owner, prefix, first_index, layers, glu_weight=False):
    for offset, layer in enumerate(layers):
        tag = "%s%d" % (prefix, first_index + offset)
        if isinstance(layer, Link):
            setattr(owner, tag, layer)
        elif glu_weight and isinstance(layer, GLU):
            setattr(owner, tag, layer.W)
        elif isinstance(layer, Residual):
            for inner, sub in enumerate(layer.layers):
                if isinstance(sub, Link):
                    setattr(owner, "%s_%d" % (tag, inner), sub)


class Stream(Chain):
    def __init__(self, *layers):
        super(Stream, self).__init__()
        self.layers = []
        if layers:
            self.layer(*layers)

    def layer(self, *layers):
        _register_layers(self, "layer_", len(self.layers), layers, glu_weight=True)
        self.layers += layers

    def __call__(self, x):
        return _apply_layers(self.layers, x)


class Module(Chain):
    def __init__(self, *layers):
        super(Module, self).__init__()
        self.layers, self.blocks = [], []
        self._links, self._submodules = [], []
        self._locked = False
        if layers:
            self.add(*layers)

    def add(self, *layers):
        _register_layers(self, "_sequential_", len(self.layers), layers)
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
        """asr/nn/nn.py:394-399 (temporary file + rename), the reference's parameter paths: asr/serializers.py"""
        from .. import serializers
        serializers.save(filename, self)

    def load(self, filename):
        """asr/nn/nn.py:401-406"""
        if os.path.isfile(filename):
            print("Loading {} ...".format(filename))
            from .. import serializers
            serializers.load(filename, self)
            return True
        return False

    def __call__(self, x):
        return _apply_layers(self.layers, x)
