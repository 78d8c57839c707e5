"""GRU links.  They reach the reference's ``asr.nn`` namespace through ``from chainer.links import *``
(asr/nn/nn.py:3); the reference never instantiates one, so the gate convention is cuDNN's / torch.nn.GRU's
(SURVEY.md section 8, row a17).  Sequences use the reference's (B, D, T) layout (as nn.SRU does)."""
import math

import torch

from .. import functions
from ..link import Link, Parameter, Uniform


class _GRUBase(Link):
    def __init__(self, in_size, out_size, ndir):
        super().__init__()
        self.in_size, self.out_size, self.ndir = in_size, out_size, ndir
        k = 1.0 / math.sqrt(out_size)
        init = Uniform(k)
        self.w_ih = Parameter()
        self.w_hh = Parameter(init((ndir, 3 * out_size, out_size)))
        self.b_ih = Parameter(init((ndir, 3 * out_size)))
        self.b_hh = Parameter(init((ndir, 3 * out_size)))
        self._init = init
        if in_size is not None:
            self._initialize_params(in_size)

    def _initialize_params(self, in_size):
        self.in_size = in_size
        self.w_ih.data = self._init((self.ndir, 3 * self.out_size, in_size)).to(self.w_ih.device)

    def __call__(self, x):
        """x (B, D, T) -> (B, H, T); bidirectional outputs are summed (Deep-Speech-2 style)."""
        if self.w_ih.numel() == 0:
            self._initialize_params(x.shape[1])
        return functions.gru(x, self.w_ih, self.w_hh, self.b_ih, self.b_hh, self, self.ndir)


class GRU(_GRUBase):
    def __init__(self, in_size, out_size=None):
        if out_size is None:
            in_size, out_size = None, in_size
        super().__init__(in_size, out_size, 1)


class BiGRU(_GRUBase):
    def __init__(self, in_size, out_size=None):
        if out_size is None:
            in_size, out_size = None, in_size
        super().__init__(in_size, out_size, 2)


class _NStep(Link):
    def __init__(self, n_layers, in_size, out_size, dropout, ndir):
        super().__init__()
        self.n_layers = n_layers
        self.dropout = dropout
        for i in range(n_layers):
            setattr(self, "l%d" % i, _GRUBase(in_size if i == 0 else out_size, out_size, ndir))

    def __call__(self, x):
        for i in range(self.n_layers):
            x = getattr(self, "l%d" % i)(x)
            if self.dropout and i + 1 < self.n_layers:
                x = functions.dropout(x, self.dropout)
        return x


class NStepGRU(_NStep):
    def __init__(self, n_layers, in_size, out_size, dropout=0):
        super().__init__(n_layers, in_size, out_size, dropout, 1)


class NStepBiGRU(_NStep):
    def __init__(self, n_layers, in_size, out_size, dropout=0):
        super().__init__(n_layers, in_size, out_size, dropout, 2)
