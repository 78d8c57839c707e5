"""nn.Convolution1D -- asr/nn/convolution_1d.py:7-38: ksize 1, stride 1, pad 0 over (B, C, T)."""
from .. import functions
from ..link import Link, Parameter, get_initializer


class Convolution1D(Link):
    def __init__(self, in_channels, out_channels, nobias=False, initialW=None, initial_bias=None, cover_all=False):
        super().__init__()
        self.out_channels = out_channels
        self.stride = 1
        self.pad = 0
        self.cover_all = cover_all
        self.initialW = initialW
        self.ksize = (1,)
        self.output_float32 = False
        self.W = Parameter()
        if in_channels is not None:
            self._initialize_params(in_channels)
        if nobias:
            self.b = None
        else:
            self.b = Parameter(get_initializer(0 if initial_bias is None else initial_bias)((out_channels,)))

    def _initialize_params(self, in_channels):
        self.in_channels = in_channels
        self.W.data = get_initializer(self.initialW)((self.out_channels, in_channels) + self.ksize).to(self.W.device)

    def __call__(self, x):
        if self.W.numel() == 0:
            self._initialize_params(x.shape[1])
        return functions.convolution_1d(x, self.W, self.b, self, self.output_float32)
