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

    def __call__(self, x, x_length=None, hx=None):
        """x (B, D, T) -> (B, H, T); bidirectional outputs are summed (Deep-Speech-2 style).  x_length (B) int32 on the device:
        run every utterance over its own length (NStepBiGRU semantics on the padded block; functions.gru).
        hx (ndir, B, H) float32: an initial state -- the call then returns (y, hy), hy (ndir, B, H) the final state, so that a long
        sequence can be fed in pieces (the way the reference's SRU model carries its contexts, run/ctc/sru/model.py:105-122)."""
        if self.w_ih.numel() == 0:
            self._initialize_params(x.shape[1])
        return functions.gru(x, self.w_ih, self.w_hh, self.b_ih, self.b_hh, self, self.ndir, x_length, hx)


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
    """chainer.links.NStepGRU / NStepBiGRU call semantics: ``hy, ys = rnn(hx, xs)`` with ``xs`` a list of (T_i, I) sequences
    (any lengths), ``hx`` None (zero initial state) or (n_layers * ndir, B, H) (a given one runs the layer on the per-step kernels),
    ``ys`` a list of (T_i, ndir * H) outputs with the two directions CONCATENATED, ``hy`` (n_layers * ndir, B, H) the last
    states.  The reference never builds one, so this is API surface, not hot path: sequences are grouped by
    length, every group runs as one batch through per-direction GRU links (the backward direction on the time-reversed
    group); joins / reversals are torch copies."""

    def __init__(self, n_layers, in_size, out_size, dropout, ndir):
        super().__init__()
        self.n_layers, self.out_size, self.dropout, self.ndir = n_layers, out_size, dropout, ndir
        for i in range(n_layers):
            for d in range(ndir):
                setattr(self, "l%d_%d" % (i, d), _GRUBase(in_size if i == 0 else out_size * ndir, out_size, 1))

    def __call__(self, hx, xs):
        if not isinstance(xs, (list, tuple)) or len(xs) == 0:
            raise TypeError("xs must be a list of (T_i, I) sequences")
        if hx is not None and tuple(hx.shape) != (self.n_layers * self.ndir, len(xs), self.out_size):
            raise ValueError("hx must have shape (n_layers * ndir, B, H) = %s, got %s"
                             % ((self.n_layers * self.ndir, len(xs), self.out_size), tuple(hx.shape)))
        groups = {}
        for idx, x in enumerate(xs):
            groups.setdefault(int(x.shape[0]), []).append(idx)
        ys = [None] * len(xs)
        hy = [[None] * len(xs) for _ in range(self.n_layers * self.ndir)]
        for T, members in groups.items():
            h = torch.stack([xs[i] for i in members], dim=0).permute(0, 2, 1)        # (Bg, I, T), the links' layout
            for layer in range(self.n_layers):
                outs = []
                for d in range(self.ndir):
                    link = getattr(self, "l%d_%d" % (layer, d))
                    hin = h if d == 0 else torch.flip(h, dims=(2,))
                    if hx is None:
                        y = link(hin)
                    else:
                        h0 = hx[layer * self.ndir + d][torch.as_tensor(members, device=hx.device)].unsqueeze(0)
                        y, hlast = link(hin, hx=h0.to(torch.float32).contiguous())
                    y = y if d == 0 else torch.flip(y, dims=(2,))
                    outs.append(y)
                    last = (y[:, :, -1] if d == 0 else y[:, :, 0]).float() if hx is None else hlast[0]
                    for k, i in enumerate(members):
                        hy[layer * self.ndir + d][i] = last[k]
                h = outs[0] if self.ndir == 1 else torch.cat(outs, dim=1)
                if self.dropout and layer + 1 < self.n_layers:
                    h = functions.dropout(h, self.dropout)
            for k, i in enumerate(members):
                ys[i] = h[k].permute(1, 0)                                              # (T_i, ndir * H)
        return torch.stack([torch.stack(row, dim=0) for row in hy], dim=0), ys


class NStepGRU(_NStep):
    def __init__(self, n_layers, in_size, out_size, dropout=0):
        super().__init__(n_layers, in_size, out_size, dropout, 1)


class NStepBiGRU(_NStep):
    def __init__(self, n_layers, in_size, out_size, dropout=0):
        super().__init__(n_layers, in_size, out_size, dropout, 2)
