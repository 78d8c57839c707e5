"""nn.SRU -- asr/nn/sru.py:229-476 (SRUFunction, sru(), SRU link) on the HIP path.

Same parameters (W (3D, D) rows [z; f; r], B (2D,) = [b_f; b_r]), same call signature
``sru(x, initial_ct, mask_x=None) -> (H, C, C[..., -1])`` on (B, D, T) arrays, same unscaled {0,1} dropout mask per
(batch, feature) (asr/nn/sru.py:473-476).  Unlike the reference, a masked input is not modified in place.
"""
import torch

from .. import functions, _ops
from ..link import Link, Parameter, get_initializer, grad_buffer, grads_queued

BF16, F32 = _ops.BF16, torch.float32


class SRUFunction(torch.autograd.Function):
    """Physical tensors: x (T, B, D) bf16, c0 (B, D) f32, mask (B, D) f32 or None."""

    @staticmethod
    def forward(ctx, x, W, Bias, c0, mask, w16, w16t, use_tanh):
        ctx.set_materialize_grads(False)
        T, Bn, D = x.shape
        xm = x if mask is None else _ops.sru_combine(x, None, mask)
        U = _ops.gemm_nt(xm.reshape(T * Bn, D), w16, None, F32)                  # asr/nn/sru.py:340-341
        H, C, cT = _ops.sru_fwd(x, U, Bias.detach(), c0.detach(), mask, use_tanh)
        ctx.save_for_backward(x, xm, U, C, c0.detach(), w16t)
        ctx.mask = mask
        ctx.params = (W, Bias)
        ctx.meta = (use_tanh, ctx.needs_input_grad[0], ctx.needs_input_grad[3])
        return H, C, cT

    @staticmethod
    def backward(ctx, gH, gC, gcT):
        x, xm, U, C, c0, w16t = ctx.saved_tensors
        W, Bias = ctx.params
        use_tanh, need_dx, need_dc0 = ctx.meta
        if gC is not None:
            raise NotImplementedError("gradients through the full cell sequence are not defined by the reference either")
        T, Bn, D = x.shape
        gU, gxh, gc0 = _ops.sru_bwd(x, U, Bias.detach(), C, c0, ctx.mask, gH.contiguous() if gH is not None else None,
                                    gcT.contiguous() if gcT is not None else None, grad_buffer(Bias), use_tanh)
        _ops.gemm_tn_acc(gU, xm.reshape(T * Bn, D), grad_buffer(W))                # asr/nn/sru.py:429
        grads_queued(W, Bias)
        gx = None
        if need_dx:
            gproj = _ops.gemm_nt(gU, w16t, None, BF16).reshape(T, Bn, D)           # asr/nn/sru.py:421-422
            gx = _ops.sru_combine(gproj, gxh, ctx.mask)                            # + highway, * mask (:422-425)
        return gx, None, None, (gc0 if need_dc0 else None), None, None, None, None


def sru(x, W, B, initial_ct, use_tanh=True, mask_x=None, link=None):
    """x (B, D, T) -> (H (B, D, T), C (B, D, T), c_T (B, D))   (asr/nn/sru.py:435-439)."""
    link = link if link is not None else _DEFAULT_LINK
    p = functions.phys3(x)
    w16 = link.compute_copy("w16", W, lambda w: _ops.cast_bf16(w), "plain")
    w16t = link.compute_copy("w16t", W, lambda w: _ops.cast_bf16(w, transpose=True), "t_first")
    H, C, cT = SRUFunction.apply(p, W, B, initial_ct, mask_x, w16, w16t, bool(use_tanh))
    return functions.logical3(H), functions.logical3(C), cT


_DEFAULT_LINK = Link()


class SRU(Link):
    def __init__(self, channels, use_tanh=True, dropout=0, initialW=None, initial_bias=0):
        super().__init__()
        self.channels = channels
        self.use_tanh = use_tanh
        self.dropout = dropout
        self._initialW, self._initial_bias = initialW, initial_bias
        self.W = Parameter()
        self.B = Parameter()
        if channels is not None:
            self._initialize_params(channels)

    def _initialize_params(self, channels):
        self.channels = channels
        self.W.data = get_initializer(self._initialW)((channels * 3, channels)).to(self.W.device)
        self.B.data = get_initializer(self._initial_bias)((channels * 2,)).to(self.B.device)

    def __call__(self, x, initial_ct, mask_x=None):
        if self.W.numel() == 0:
            self._initialize_params(x.shape[1])
        batchsize, feature_dimension = x.shape[:2]
        if initial_ct is None:
            initial_ct = torch.empty((batchsize, feature_dimension), dtype=F32, device=x.device)
            _ops.fill_(initial_ct, 0.0)
        if self.dropout == 0 or not functions.train_mode[0]:
            return sru(x, self.W, self.B, initial_ct, self.use_tanh, link=self)
        mask_x = self.generate_dropout_mask(x) if mask_x is None else mask_x
        return sru(x, self.W, self.B, initial_ct, self.use_tanh, mask_x, link=self)

    def generate_dropout_mask(self, x):
        # host-side RNG, (B, D) values only (asr/nn/sru.py:473-476)
        mask = torch.rand(x.shape[0], x.shape[1]) >= self.dropout
        return mask.to(torch.float32).to(x.device)
