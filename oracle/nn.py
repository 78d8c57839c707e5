"""CPU oracle for the operator layer (asr/nn).  TEST INFRASTRUCTURE -- see oracle/__init__.py.

numpy restatements of arithmetic the reference owns (layer-norm, SRU, weight-norm), torch-CPU fp32 for what
the reference delegates to Chainer (conv, max-pool, maxout, GRU; PARITY UNPINNED at that boundary).
All functions use the reference's logical layouts: images (B, C, H, T), sequences (B, D, T).
"""
import numpy as np
import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------- conv / pooling
def conv2d_causal(x, W, b=None, pad_h=0):
    """nn.Convolution2D(pad=(pad_h, kw-1)) followed by x[..., :-(kw-1)]  (asr/nn/nn.py:235-238,
    run/ctc/cnn/model.py:43-44).  x (B,C,H,T), W (Co,Ci,kh,kw) torch tensors."""
    kw = W.shape[3]
    y = F.conv2d(x, W, b, stride=1, padding=(pad_h, kw - 1))
    return y[..., :-(kw - 1)] if kw > 1 else y


def maxout2(x):
    """chainer.functions.maxout(x, 2, axis=1): max over adjacent channel pairs (asr/nn/nn.py:45-50)."""
    s = x.shape
    return x.reshape(s[0], s[1] // 2, 2, *s[2:]).max(dim=2)[0]


def maxpool_h(x, k):
    """chainer.functions.max_pooling_2d(x, (k,1)) with its defaults stride=ksize, cover_all=True (asr/nn/nn.py:95-103)."""
    return F.max_pool2d(x, kernel_size=(k, 1), stride=(k, 1), ceil_mode=True)


# ---------------------------------------------------------------------------------------------- layer norm
def normalize_layer_fwd(x):
    """asr/nn/layernorm.py:33-48 (no epsilon).  x (B, C, H, T) or (B, V, T) numpy."""
    size = x.shape[1] * x.shape[2]
    mean = x.mean(axis=(1, 2), keepdims=True)
    diff = x - mean
    std = np.sqrt((diff ** 2).sum(axis=(1, 2), keepdims=True) / size)
    return diff / std, diff, std


def normalize_layer_bwd(gy, diff, std):
    """asr/nn/layernorm.py:50-61 restated (sum-to-shape then broadcast)."""
    size = diff.shape[1] * diff.shape[2]
    std_grad = (-gy / (std ** 2) * diff).sum(axis=(1, 2), keepdims=True)
    var_grad = np.broadcast_to(std_grad * 0.5 / std, diff.shape) / size
    var_grad = var_grad * 2 * diff
    var_grad = var_grad + gy / std
    grad_broad = (-var_grad).sum(axis=(1, 2), keepdims=True)
    mean_grad = np.broadcast_to(grad_broad, diff.shape) / size
    return var_grad + mean_grad


def layer_normalization(x, gamma, beta):
    """nn.LayerNormalization.__call__ (asr/nn/nn.py:260-265): normalise, scale and bias along axis 1."""
    xh, diff, std = normalize_layer_fwd(x)
    shape = [1, -1] + [1] * (x.ndim - 2)
    return xh * gamma.reshape(shape) + beta.reshape(shape), (xh, diff, std)


def layer_normalization_bwd(gy, gamma, cache):
    xh, diff, std = cache
    shape = [1, -1] + [1] * (gy.ndim - 2)
    axes = tuple(i for i in range(gy.ndim) if i != 1)
    dgamma = (gy * xh).sum(axis=axes)
    dbeta = gy.sum(axis=axes)
    dx = normalize_layer_bwd(gy * gamma.reshape(shape), diff, std)
    return dx, dgamma, dbeta


# ---------------------------------------------------------------------------------------------- SRU
def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def sru_fwd(X, W, Bias, c0, use_tanh=True, mask=None):
    """asr/nn/sru.py:289-324 (forward_cpu) == kernel K1 (:17-73).  X (B,D,T), W (3D,D) rows [z;f;r], Bias (2D,)
    = [b_f; b_r].  The highway term uses x * mask; U = W X uses the caller's X (forward_cpu does not mask it)."""
    Bn, D, T = X.shape
    U = np.matmul(W, X)
    Z, Fg, R = U[:, :D], U[:, D:2 * D], U[:, 2 * D:]
    xm = X if mask is None else X * mask[..., None]
    C = np.empty_like(X)
    H = np.empty_like(X)
    c = c0
    for t in range(T):
        f = _sigmoid(Fg[..., t] + Bias[:D])
        r = _sigmoid(R[..., t] + Bias[D:])
        c = f * c + (1 - f) * Z[..., t]
        g = np.tanh(c) if use_tanh else c
        H[..., t] = r * g + (1 - r) * xm[..., t]
        C[..., t] = c
    return H, C, C[..., -1]


def sru_bwd(X, W, Bias, c0, gH, gcT, use_tanh=True):
    """Kernel K2 (asr/nn/sru.py:75-191) + host part (:421-433) restated in numpy (no dropout mask)."""
    Bn, D, T = X.shape
    U = np.matmul(W, X)
    Z, Fg, R = U[:, :D], U[:, D:2 * D], U[:, 2 * D:]
    _, C, _ = sru_fwd(X, W, Bias, c0, use_tanh)
    gU = np.zeros_like(U)
    gxh = np.zeros_like(X)
    gb = np.zeros(2 * D, dtype=X.dtype)
    gc = np.zeros_like(c0) if gcT is None else gcT.copy()
    for t in range(T - 1, -1, -1):
        f = _sigmoid(Fg[..., t] + Bias[:D])
        r = _sigmoid(R[..., t] + Bias[D:])
        c = C[..., t]
        cp = c0 if t == 0 else C[..., t - 1]
        g = np.tanh(c) if use_tanh else c
        gh = gH[..., t]
        gbr = gh * (g - X[..., t]) * (1 - r) * r
        gtanh = (1 - g * g) if use_tanh else 1.0
        gct = gh * r * gtanh
        gbf = (gct + gc) * (cp - Z[..., t]) * (1 - f) * f
        gxh[..., t] = gh * (1 - r)
        gU[:, :D, t] = (gct + gc) * (1 - f)
        gU[:, D:2 * D, t] = gbf
        gU[:, 2 * D:, t] = gbr
        gb[:D] += gbf.sum(axis=0)
        gb[D:] += gbr.sum(axis=0)
        gc = (gct + gc) * f
    gX = np.einsum("od,bot->bdt", W, gU) + gxh
    gW = np.einsum("bot,bdt->od", gU, X)
    return gX, gW, gb, gc


# ---------------------------------------------------------------------------------------------- weight norm
def weightnorm_W(V, g):
    """asr/nn/convolution_2d.py:21-25,62-64: W = g * V / (||V|| + 1e-9), norm per output channel."""
    norm = np.sqrt((V ** 2).sum(axis=(1, 2, 3), keepdims=True)) + 1e-9
    return g * V / norm, V / norm, norm


def weightnorm_bwd(gW, V, g):
    """asr/nn/convolution_2d.py:92-93."""
    _, Vn, norm = weightnorm_W(V, g)
    gg = (gW * Vn).sum(axis=(1, 2, 3), keepdims=True)
    gV = g * (gW - gg * Vn) / norm
    return gV, gg


# ---------------------------------------------------------------------------------------------- GRU (torch CPU)
def bigru_sum(x_tbi, params, H, ndir=2):
    """torch.nn.GRU on CPU; directions summed (the DS2-style stack of BASELINE.json).  params: dict with
    w_ih (ndir,3H,I), w_hh (ndir,3H,H), b_ih (ndir,3H), b_hh (ndir,3H) torch tensors.  x (T,B,I)."""
    I = x_tbi.shape[2]
    gru = torch.nn.GRU(I, H, num_layers=1, bidirectional=(ndir == 2))
    with torch.no_grad():
        for d, suf in enumerate(["", "_reverse"][:ndir]):
            getattr(gru, "weight_ih_l0" + suf).copy_(params["w_ih"][d])
            getattr(gru, "weight_hh_l0" + suf).copy_(params["w_hh"][d])
            getattr(gru, "bias_ih_l0" + suf).copy_(params["b_ih"][d])
            getattr(gru, "bias_hh_l0" + suf).copy_(params["b_hh"][d])
    out, _ = gru(x_tbi)
    if ndir == 2:
        out = out[..., :H] + out[..., H:]
    return out, gru


# ---------------------------------------------------------------------------------------------- optimiser
def clip_decay_adam(p, g, m, v, step, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, decay=1e-5, clip=1.0):
    """GradientClipping(clip) -> WeightDecay(decay) -> Chainer-v2 Adam (run/ctc/cnn/train.py:142-147; Chainer is an
    absent third-party: formulas recalled from chainer/optimizers/adam.py v2, PARITY UNPINNED)."""
    norm = np.sqrt((g.astype(np.float64) ** 2).sum())
    rate = clip / norm
    if clip > 0 and rate < 1:
        g = g * rate
    g = g + decay * p
    m = m + (1 - beta1) * (g - m)
    v = v + (1 - beta2) * (g * g - v)
    lr = alpha * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    p = p - lr * m / (np.sqrt(v) + eps)
    return p, m, v


def batch_normalization(x, gamma, beta, avg_mean, avg_var, eps=2e-5, decay=0.9, train=True):
    """chainer.links.BatchNormalization as documented by Chainer (third party, absent: parity unpinned; torch's
    batch_norm is the cross-check in tests): x (B, C, ...) float64 numpy; returns y and the updated running averages."""
    import numpy as np
    axes = (0,) + tuple(range(2, x.ndim))
    shape = (1, -1) + (1,) * (x.ndim - 2)
    if train:
        mean, var = x.mean(axis=axes), x.var(axis=axes)
        n = x.size // x.shape[1]
        avg_mean = decay * avg_mean + (1 - decay) * mean
        avg_var = decay * avg_var + (1 - decay) * var * n / max(n - 1, 1)
    else:
        mean, var = avg_mean, avg_var
    y = gamma.reshape(shape) * (x - mean.reshape(shape)) / np.sqrt(var.reshape(shape) + eps) + beta.reshape(shape)
    return y, avg_mean, avg_var
