"""nn.Convolution1D: the per-frame affine map of the reference (asr/nn/convolution_1d.py:7-38: a ConvolutionND with
kernel 1, stride 1, no padding over (B, C, T)), i.e. one GEMM over the (T*B, C) rows of the physical layout.

The weight keeps Chainer's (out, in, 1) shape; when ``in_channels`` is None it is sized from the first input, as there.
"""
from .. import functions
from ..link import Link, Parameter, get_initializer

_FIXED = dict(stride=1, pad=0, ksize=(1,))          # what the reference hard-codes


class Convolution1D(Link):
    def __init__(self, in_channels, out_channels, nobias=False, initialW=None, initial_bias=None, cover_all=False):
        super().__init__()
        vars(self).update(_FIXED, out_channels=out_channels, cover_all=cover_all, initialW=initialW, in_channels=None,
                          output_float32=False)      # output_float32: set by the acoustic model on the logit layers
        self.W = Parameter()
        self.b = None if nobias else Parameter(get_initializer(initial_bias if initial_bias is not None else 0)((out_channels,)))
        if in_channels is not None:
            self._initialize_params(in_channels)

    def _initialize_params(self, in_channels):
        shape = (self.out_channels, in_channels) + self.ksize
        self.W.data = get_initializer(self.initialW)(shape).to(self.W.device)
        self.in_channels = in_channels

    def __call__(self, x):
        if self.W.numel() == 0:         # lazily sized from x (B, C, T)
            self._initialize_params(x.shape[1])
        return functions.convolution_1d(x, self.W, self.b, self, self.output_float32)
