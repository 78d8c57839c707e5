"""Convolution2D links.

``PlainConvolution2D``   chainer.links.Convolution2D as the reference uses it (asr/nn/nn.py:238): stride 1.
``Convolution2D``        the weight-normalised link of asr/nn/convolution_2d.py:127-189:  W = g * V / (||V|| + 1e-9),
                         data-dependent initialisation of g and b on the first call (:177-187).
"""
import torch

from .. import functions, _ops
from ..link import Link, Parameter, get_initializer, grad_buffer, grads_queued


def _pair(x):
    if hasattr(x, "__getitem__"):
        return tuple(x)
    return x, x


class PlainConvolution2D(Link):
    def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False, initialW=None,
                 initial_bias=None, causal=False):
        super().__init__()
        if ksize is None:
            out_channels, ksize, in_channels = in_channels, out_channels, None
        self.ksize = _pair(ksize)
        self.stride = (1, 1)
        self.pad = _pair(pad)
        self.out_channels = out_channels
        self.causal = causal
        self.output_float32 = False
        self._initialW = initialW
        self.W = Parameter()
        self.b = None if nobias else Parameter(get_initializer(0 if initial_bias is None else initial_bias)((out_channels,)))
        if in_channels is not None:
            self._initialize_params(in_channels)

    def _initialize_params(self, in_channels):
        kh, kw = self.ksize
        self.W.data = get_initializer(self._initialW)((self.out_channels, in_channels, kh, kw)).to(self.W.device)

    def __call__(self, x):
        if self.W.numel() == 0:
            self._initialize_params(x.shape[1])
        return functions.convolution_2d(x, self.W, self.b, self, self.pad, self.causal, self.output_float32)

    def fused_maxout_pool(self, x, k):
        """this layer, Maxout(2) and MaxPooling2D((k, 1)) as one pass where the layer is a model's first (functions.convolution_maxout_pool);
        None: run the three layers"""
        if self.output_float32:
            return None
        if self.W.numel() == 0:
            self._initialize_params(x.shape[1])
        return functions.convolution_maxout_pool(x, self.W, self.b, self, self.pad, self.causal, k)


class _WeightNorm(torch.autograd.Function):
    """W = g * V / (||V|| + 1e-9) and its gradient (asr/nn/convolution_2d.py:21-25,62-64,92-93)."""

    @staticmethod
    def forward(ctx, V, g):
        W, norm = _ops.weightnorm_fwd(V.detach(), g.detach())
        ctx.save_for_backward(norm)
        ctx.params = (V, g)
        return W

    @staticmethod
    def backward(ctx, gW):
        (norm,) = ctx.saved_tensors
        V, g = ctx.params
        _ops.weightnorm_bwd(gW.contiguous(), V.detach(), g.detach(), norm, grad_buffer(V), grad_buffer(g))
        grads_queued(V, g)
        return None, None


class Convolution2D(Link):
    def __init__(self, in_channels, out_channels, ksize=None, stride=1, pad=0, nobias=False, initialV=None,
                 causal=False, **kwargs):
        super().__init__()
        if ksize is None:
            out_channels, ksize, in_channels = in_channels, out_channels, None
        self.ksize = _pair(ksize)
        self.stride = (1, 1)
        self.pad = _pair(pad)
        self.out_channels = out_channels
        self.nobias = nobias
        self.causal = causal
        self.output_float32 = False
        self._initialV = initialV
        self.V = Parameter()
        if in_channels is not None:
            self._initialize_V(in_channels)
        self.b = None if nobias else Parameter()
        self.g = Parameter()

    def _initialize_V(self, in_channels):
        kh, kw = self.ksize
        self.V.data = get_initializer(self._initialV)((self.out_channels, in_channels, kh, kw)).to(self.V.device)

    @property
    def W(self):
        with torch.no_grad():
            return _ops.weightnorm_fwd(self.V.detach(), self.g.detach())[0]

    def __call__(self, x):
        if self.g.numel() == 0:
            # data-dependent initialisation (asr/nn/convolution_2d.py:177-187): run with g = 1, no bias; set
            # g = 1/std_t, b = -mean_t/std_t per output channel; return the normalised output of this first call.
            if self.V.numel() == 0:
                self._initialize_V(x.shape[1])
            dev = self.V.device
            with torch.no_grad():
                ones = torch.empty((self.out_channels, 1, 1, 1), dtype=torch.float32, device=dev)
                _ops.fill_(ones, 1.0)
                W, _ = _ops.weightnorm_fwd(self.V.detach(), ones)
                helper = Link()
                t = functions.convolution_2d(x, W, None, helper, self.pad, self.causal, True)     # f32 output
                tp = t.permute(3, 0, 2, 1).contiguous()                                              # (T, B, H, Co), no-op
                mean, std = _ops.channel_mean_std(tp.reshape(-1, self.out_channels))
                g, b = _ops.weightnorm_init(mean, std)
                self.g.data = g.reshape(self.out_channels, 1, 1, 1)
                if not self.nobias:
                    self.b.data = b
                y = _ops.channel_affine(tp.reshape(-1, self.out_channels), g, b).reshape(tp.shape)
            return y.permute(1, 3, 2, 0)
        W = _WeightNorm.apply(self.V, self.g)
        return functions.convolution_2d_given_weight(x, W, self.b, self, self.pad, self.causal, self.output_float32)

    def fused_maxout_pool(self, x, k):
        """as PlainConvolution2D.fused_maxout_pool; the call that initialises g and b from the data runs the plain layers"""
        if self.output_float32 or self.g.numel() == 0 or self.V.numel() == 0:
            return None
        if not functions.convolution_maxout_pool_ok(x, self.V.shape, self.pad, self.causal, k):
            return None
        W = _WeightNorm.apply(self.V, self.g)
        return functions.convolution_maxout_pool(x, W, self.b, self, self.pad, self.causal, k, given_weight=True)
