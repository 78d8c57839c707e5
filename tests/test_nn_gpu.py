"""asr.nn / asr.fft operators on the HIP path against the CPU oracle (reference layouts in, reference layouts out)."""
import os

import numpy as np
import pytest
import torch

from asr._lib import act_dtype as _act_dtype      # bfloat16, or float16 when the half build is under test (ASR_ACT=f16)

from oracle import fft as offt
from oracle import nn as onn

pytestmark = pytest.mark.gpu
BF16, F32 = _act_dtype(), torch.float32


def _bf(t):
    return t.to(BF16).to(F32)


def _rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


# ------------------------------------------------------------------------------------------------ features
def test_logfbank_batch_matches_oracle(device):
    from asr import fft
    rs = np.random.RandomState(0)
    sigs = [np.round(rs.randn(n) * 3000).astype(np.int16) for n in (16000, 9000, 12345, 700)]
    proc = fft.Processor(device=device)
    x, xl = proc.logfbank_batch(sigs)
    xr, xlr = offt.logfbank_minibatch(sigs)
    assert x.shape == xr.shape and x.dtype == F32
    assert np.array_equal(xl.cpu().numpy(), xlr)
    np.testing.assert_allclose(x.cpu().numpy(), xr, rtol=0, atol=2e-4)          # log-mel ~ 13 +- 1, f32 FFT vs f64
    mean = rs.randn(3, 40).astype(np.float32)
    std = (rs.rand(3, 40) + 0.5).astype(np.float32)
    xn, _ = proc.logfbank_batch(sigs, mean, std)
    xrn, _ = offt.logfbank_minibatch(sigs, mean=mean, std=std)
    np.testing.assert_allclose(xn.cpu().numpy(), xrn, rtol=0, atol=5e-4)


def test_fft_functions_match_reference_goldens(device, golden_dir):
    """the reference's own outputs (asr/fft.py on a seeded power spectrum) through the device functions"""
    from asr import fft
    g = np.load(os.path.join(golden_dir, "fft.npz"))
    assert np.array_equal(fft.get_filterbanks(40, 512, 16000), g["fbank"])
    pspec = torch.tensor(g["pspec"], dtype=F32, device=device)
    lm = fft.compute_logmel(pspec, 16000, fbank=g["fbank"], nfft=512, nfilt=40)
    np.testing.assert_allclose(lm.cpu().numpy(), g["logmel_full"], atol=2e-5)
    a, d, dd = fft.compute_deltas(lm)
    np.testing.assert_allclose(a.cpu().numpy(), g["logmel"], atol=2e-5)
    np.testing.assert_allclose(d.cpu().numpy(), g["delta"], atol=2e-5)
    np.testing.assert_allclose(dd.cpu().numpy(), g["delta_delta"], atol=2e-5)
    np.testing.assert_allclose(fft.compute_delta(lm).cpu().numpy(), offt.compute_delta(g["logmel_full"]), atol=2e-5)
    sig = np.round(np.random.RandomState(3).randn(4000) * 2000).astype(np.int16)
    ps = fft.get_specgram(torch.tensor(sig, device=device), 16000, 0.032, 0.01, 512, 0.97, np.hanning)
    ref = offt.get_specgram(sig, 16000, 0.032, 0.01, 512, 0.97, np.hanning)
    np.testing.assert_allclose(ps.cpu().numpy(), ref, rtol=2e-4, atol=ref.max() * 1e-6)


def test_full_size_feature_batch(device):
    """BASELINE shape: 160672 samples -> 1002 frames -> T = 1000."""
    from asr import fft
    rs = np.random.RandomState(1)
    sigs = [np.round(rs.randn(160672) * 3000).astype(np.int16) for _ in range(4)]
    x, xl = fft.Processor(device=device).logfbank_batch(sigs)
    assert x.shape == (4, 3, 40, 1000) and (xl.cpu().numpy() == 1000).all()
    xr, _ = offt.logfbank_minibatch(sigs[:1])
    # float32 transform against the float64 oracle: the lowest mel bands sit on bins that pre-emphasis has taken down to 1e-3 of the
    # spectrum's level, so in a few frames their logarithm carries the transform's rounding relative to the LARGE bins -- a heavy tail on a
    # tiny error (round 5's radix-4 kernel, this seed: rms 2.5e-6 over the 120,000 values, 7 of them beyond 1e-4, one -- mel 0 of one
    # frame -- at 3.1e-4; the radix-2 kernel it replaced drew 1.9e-4 as its largest).  The short signals above keep the 2e-4 bar.
    d = np.abs(x[0].cpu().numpy() - xr[0])
    assert d.max() <= 5e-4 and (d > 2e-4).sum() <= 3 and float(np.sqrt((d ** 2).mean())) <= 3e-5


def test_mel_band_table_and_its_fallbacks(device):
    """asr_specgram_bands with the band table (the shipped path), without one (dense rows read from memory) and with a matrix the table
    cannot hold (80 filters > 64): the same log-mel values to the last bits of the float32 sums, and the one-workgroup-per-frame kernels (ASR_DEBUG fbank_fast=0's path, reached here through a frame
    step that is not a multiple of 4) agree with them to float32 rounding"""
    from asr import fft
    rs = np.random.RandomState(5)
    B, N = 3, 12000
    sig = torch.from_numpy(np.round(rs.randn(B, N) * 3000).astype(np.int16)).to(device)
    lens = torch.tensor([N, 9000, 4321], dtype=torch.int32, device=device)
    window = torch.tensor(np.hanning(512), dtype=F32, device=device)
    for nfilt in (40, 80):
        fb = torch.tensor(fft.get_filterbanks(nfilt, 512, 16000), dtype=F32, device=device)
        frames = [fft.num_frames(int(n), 512, 160) for n in lens.cpu()]
        nfr = torch.tensor(frames, dtype=torch.int32, device=device)
        bands = fft.mel_bands(fb)
        table = bands.cpu()
        assert (int(table[64]) >= 0) == (nfilt <= 64)                       # len8[0] = -1: "did not fit"
        _, with_table = fft._specgram(sig, lens, nfr, max(frames), 512, 160, 512, 0.97, window, fb, False, bands)
        _, dense = fft._specgram(sig, lens, nfr, max(frames), 512, 160, 512, 0.97, window, fb, False, None)
        for b in range(B):
            diff = float((with_table[b, :frames[b]] - dense[b, :frames[b]]).abs().max())
            assert diff <= 4e-6, (nfilt, b, diff)      # (log-mel ~ 7 .. 20: last-bit differences of the float32 sums)
        # frame step 162 (not a multiple of 4): the general kernel; same framing arithmetic on the host
        frames2 = [fft.num_frames(int(n), 512, 162) for n in lens.cpu()]
        nfr2 = torch.tensor(frames2, dtype=torch.int32, device=device)
        ps_fast, _ = fft._specgram(sig, lens, nfr, max(frames), 512, 160, 512, 0.97, window, None, True)
        ps_gen, _ = fft._specgram(sig[:, :N - 2].contiguous(), torch.clamp(lens, max=N - 2), nfr2, max(frames2), 512, 162, 512, 0.97, window, None, True)
        assert ps_gen.shape[1] == max(frames2) and torch.isfinite(ps_gen[0, :frames2[0]]).all()
        # frame 0 starts at sample 0 in both framings: identical input, two different kernels
        np.testing.assert_allclose(ps_gen[:, 0].cpu().numpy(), ps_fast[:, 0].cpu().numpy(), rtol=2e-4, atol=float(ps_fast[:, 0].max()) * 1e-6)


# ------------------------------------------------------------------------------------------------ SRU
@pytest.mark.parametrize("name", ["tanh", "linear"])
def test_sru_forward_golden(device, golden_dir, name):
    """reference forward_cpu outputs (asr/nn/sru.py:289-324) through nn.sru on the GPU"""
    from asr.nn.sru import sru
    from asr.link import Parameter
    g = np.load(os.path.join(golden_dir, "sru.npz"))
    X, W, B, c0 = (torch.tensor(g["%s.%s" % (name, k)]) for k in ("X", "W", "B", "c0"))
    H, C, cT = sru(X.to(device), Parameter(W.to(device)), Parameter(B.to(device)), c0.to(device), bool(g[name + ".use_tanh"]))
    assert H.shape == X.shape
    # bf16 operands in the projection: compare against the oracle fed with bf16-rounded X and W, and loosely with the golden
    Hr, Cr, cTr = onn.sru_fwd(_bf(X).double().numpy(), _bf(W).double().numpy(), B.double().numpy(), c0.double().numpy(),
                              bool(g[name + ".use_tanh"]))
    assert _rel(C.cpu(), Cr) < 2e-5 and _rel(cT.cpu(), cTr) < 2e-5
    assert _rel(H.float().cpu(), Hr) < 5e-3
    assert _rel(H.float().cpu(), g[name + ".H"]) < 2e-2


@pytest.mark.parametrize("use_tanh,masked", [(True, False), (False, False), (True, True)])
def test_sru_forward_backward(device, use_tanh, masked):
    from asr.nn.sru import sru
    from asr.link import Parameter
    torch.manual_seed(0)
    Bn, D, T = 3, 64, 17
    X = _bf(torch.randn(Bn, D, T))
    W = _bf(torch.randn(3 * D, D) * 0.2)
    Bias = torch.randn(2 * D) * 0.3
    c0 = torch.randn(Bn, D)
    mask = (torch.rand(Bn, D) > 0.3).float() if masked else None
    gH = _bf(torch.randn(Bn, D, T))
    gcT = torch.randn(Bn, D)
    Xd = X.to(device).requires_grad_(True)
    Wp, Bp = Parameter(W.to(device)), Parameter(Bias.to(device))
    c0d = c0.to(device).requires_grad_(True)
    H, C, cT = sru(Xd, Wp, Bp, c0d, use_tanh, None if mask is None else mask.to(device))
    ((H.float() * gH.to(device)).sum() + (cT * gcT.to(device)).sum()).backward()
    Xm = X if mask is None else X * mask[..., None]        # GPU semantics: the projection sees the masked input (sru.py:336-341)
    Hr, Cr, cTr = onn.sru_fwd(Xm.double().numpy(), W.double().numpy(), Bias.double().numpy(), c0.double().numpy(), use_tanh)
    assert _rel(H.float().cpu(), Hr) < 5e-3 and _rel(cT.cpu(), cTr) < 1e-4
    gX, gW, gb, gc = onn.sru_bwd(Xm.double().numpy(), W.double().numpy(), Bias.double().numpy(), c0.double().numpy(),
                                 gH.double().numpy(), gcT.double().numpy(), use_tanh)
    if mask is not None:
        gX = gX * mask[..., None].numpy()
    assert _rel(Xd.grad.float().cpu(), gX) < 1e-2
    assert _rel(Wp.grad.cpu(), gW) < 1e-2
    assert _rel(Bp.grad.cpu(), gb) < 1e-2
    assert _rel(c0d.grad.cpu(), gc) < 1e-2


# ------------------------------------------------------------------------------------------------ layers
def _img(B, C, H, T, seed=0):
    return _bf(torch.randn(B, C, H, T, generator=torch.Generator().manual_seed(seed)))


@pytest.mark.parametrize("kind", ["relu", "clipped_relu", "leaky_relu", "elu", "sigmoid", "tanh", "hard_sigmoid", "softplus"])
def test_activations(device, kind):
    import asr.nn as nn
    x = _img(2, 6, 5, 7) * 3
    layer = {"relu": nn.ReLU(), "clipped_relu": nn.ClippedReLU(2.0), "leaky_relu": nn.LeakyReLU(0.1), "elu": nn.ELU(0.7),
             "sigmoid": nn.Sigmoid(), "tanh": nn.Tanh(), "hard_sigmoid": nn.HardSigmoid(), "softplus": nn.Softplus(1.5)}[kind]
    F = torch.nn.functional
    ref = {"relu": F.relu, "clipped_relu": lambda t: t.clamp(0, 2.0), "leaky_relu": lambda t: F.leaky_relu(t, 0.1),
           "elu": lambda t: F.elu(t, 0.7), "sigmoid": torch.sigmoid, "tanh": torch.tanh,
           "hard_sigmoid": lambda t: (0.2 * t + 0.5).clamp(0, 1), "softplus": lambda t: F.softplus(t, 1.5)}[kind]
    xd = x.to(device).requires_grad_(True)
    y = layer(xd)
    gy = _img(2, 6, 5, 7, 1)
    y.float().backward(gy.to(device))
    xr = x.clone().requires_grad_(True)
    yr = ref(xr)
    yr.backward(gy)
    assert y.shape == x.shape
    assert _rel(y.float().cpu(), yr.detach()) < 6e-3
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-2


def test_glu_dropout_residual_linear(device):
    import asr.nn as nn
    from asr import functions as F
    x = _img(2, 8, 5, 9)
    xd = x.to(device).requires_grad_(True)
    y = F.glu(xd)
    a, b = x[:, :4], x[:, 4:]
    assert y.shape == (2, 4, 5, 9) and _rel(y.float().cpu(), a * torch.sigmoid(b)) < 6e-3
    y.float().sum().backward()
    xr = x.clone().requires_grad_(True)
    (xr[:, :4] * torch.sigmoid(xr[:, 4:])).sum().backward()
    assert _rel(xd.grad.cpu(), xr.grad) < 1e-2
    # dropout: scaled Bernoulli mask, identical mask in backward, identity at ratio 0 and in eval mode
    d = nn.Dropout(0.4)
    xd2 = (torch.ones(4, 8, 6, 50) * 2).to(device).requires_grad_(True)
    yd = d(xd2)
    kept = (yd != 0).float().mean().item()
    assert abs(kept - 0.6) < 0.03 and torch.allclose(yd[yd != 0].float(), torch.tensor(2 / 0.6, device=device), rtol=1e-2)
    yd.float().sum().backward()
    assert torch.equal(xd2.grad != 0, yd != 0)
    assert nn.Dropout(0)(xd2) is xd2
    F.train_mode[0] = False
    assert d(xd2) is xd2
    F.train_mode[0] = True
    # residual: y = layer(x) + x  (asr/nn/nn.py:322-328)
    s = nn.Stream(nn.Residual(nn.ReLU()))
    ys = s(x.to(device))
    assert _rel(ys.float().cpu(), torch.relu(x) + x) < 6e-3
    # linear
    torch.manual_seed(1)
    lin = nn.Linear(16, 24).to_gpu()
    v = _bf(torch.randn(10, 16))
    out = lin(v.to(device))
    assert _rel(out.float().cpu(), v @ _bf(lin.W.detach().cpu()).T + lin.b.detach().cpu()) < 6e-3


def test_layernorm_3d_joint_statistics(device):
    """(B, V, T) input: the reference normalises over V and T jointly (axes 1, 2) -- asr/nn/layernorm.py:42-45."""
    import asr.nn as nn
    rs = np.random.RandomState(5)
    x = rs.uniform(-3, 3, (3, 10, 7)).astype(np.float32)
    ln = nn.LayerNormalization(10).to_gpu()
    ln.output_float32 = True
    with torch.no_grad():
        ln.gamma.copy_(torch.tensor(rs.uniform(0.5, 1.5, 10).astype(np.float32)))
        ln.beta.copy_(torch.tensor(rs.uniform(-1, 1, 10).astype(np.float32)))
    y = ln(torch.tensor(x).to(device))
    yr, _ = onn.layer_normalization(_bf(torch.tensor(x)).double().numpy(), ln.gamma.detach().cpu().double().numpy(),
                                    ln.beta.detach().cpu().double().numpy())
    assert y.shape == (3, 10, 7)
    np.testing.assert_allclose(y.detach().cpu().numpy(), yr, rtol=1e-3, atol=1e-3)


def test_weightnorm_convolution(device):
    """first call = data-dependent init returning the normalised output; later calls use W = g V / ||V||
    (asr/nn/convolution_2d.py:152-187), gradients w.r.t. V, g, b against torch autograd of the same formula."""
    import asr.nn as nn
    torch.manual_seed(0)
    conv = nn.Convolution2D(4, 6, (3, 5), pad=(1, 4), weightnorm=True).to_gpu()
    x = _img(3, 4, 7, 20, 2)
    y0 = conv(x.to(device))[..., :-4]
    V = conv.V.detach().cpu()
    Vn = V / (V.pow(2).sum(dim=(1, 2, 3), keepdim=True).sqrt() + 1e-9)
    t = onn.conv2d_causal(x, _bf(Vn), None, 1)
    tfull = torch.nn.functional.conv2d(x, _bf(Vn), None, padding=(1, 4))          # statistics are taken over the padded output
    mean, std = tfull.mean(dim=(0, 2, 3)), tfull.var(dim=(0, 2, 3), unbiased=False).sqrt()
    assert _rel(conv.g.detach().cpu().reshape(-1), 1 / std) < 5e-3
    assert _rel(conv.b.detach().cpu(), -mean / std) < 2e-2
    assert _rel(y0.float().cpu(), (t - mean[None, :, None, None]) / std[None, :, None, None]) < 1e-2
    y1 = conv(x.to(device))[..., :-4]
    assert _rel(y1.float().cpu(), y0.float().cpu()) < 1e-2          # second call reproduces the initialised output
    gy = _img(3, 6, 7, 20, 3)
    y1.float().backward(gy.to(device))
    Vr = V.clone().requires_grad_(True)
    gr = conv.g.detach().cpu().clone().requires_grad_(True)
    br = conv.b.detach().cpu().clone().requires_grad_(True)
    Wr = gr * Vr / (Vr.pow(2).sum(dim=(1, 2, 3), keepdim=True).sqrt() + 1e-9)
    onn.conv2d_causal(x, Wr, br, 1).backward(gy)
    assert _rel(conv.V.grad.cpu(), Vr.grad) < 3e-2
    assert _rel(conv.g.grad.cpu(), gr.grad) < 3e-2
    assert _rel(conv.b.grad.cpu(), br.grad) < 1e-2


def test_generic_layout_entry(device):
    """ops accept tensors in the reference's plain (B, C, H, T) float32 layout and convert once"""
    import asr.nn as nn
    x = _img(2, 8, 6, 11)
    y = nn.Maxout(2)(x.to(device))
    assert _rel(y.float().cpu(), onn.maxout2(x)) == 0
    y = nn.MaxPooling2D(ksize=(2, 1))(x.to(device))
    assert _rel(y.float().cpu(), onn.maxpool_h(x, 2)) == 0


def test_batch_normalization(device):
    """nn.BatchNormalization (chainer.links name, asr/nn/nn.py:3): forward, running averages, gradients, test mode --
    against oracle/nn.py and torch's batch_norm on the CPU (bf16 activations: 2e-2 tolerances)."""
    import asr.functions as F
    from asr import nn
    from oracle import nn as onn
    rs = np.random.RandomState(3)
    B, C, H, T = 3, 24, 5, 17
    x = (rs.randn(B, C, H, T) * 1.5 + 0.7).astype(np.float32)
    x = torch.from_numpy(x).to(BF16).float()          # representable values
    bn = nn.BatchNormalization(C)
    with torch.no_grad():
        bn.gamma.copy_(torch.from_numpy(rs.rand(C).astype(np.float32) + 0.5))
        bn.beta.copy_(torch.from_numpy(rs.randn(C).astype(np.float32) * 0.1))
    g0, b0 = bn.gamma.detach().clone().numpy().astype(np.float64), bn.beta.detach().clone().numpy().astype(np.float64)
    bn.to_gpu(0)
    xd = x.to(device).requires_grad_(True)
    y = bn(xd)
    want, am, av = onn.batch_normalization(x.numpy().astype(np.float64), g0, b0, np.zeros(C), np.ones(C))
    np.testing.assert_allclose(y.float().detach().cpu().numpy(), want, rtol=2e-2, atol=2e-2)
    np.testing.assert_allclose(bn.avg_mean.cpu().numpy(), am, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(bn.avg_var.cpu().numpy(), av, rtol=1e-3, atol=1e-5)
    # gradients against torch autograd on the CPU
    gy = torch.from_numpy(rs.randn(B, C, H, T).astype(np.float32)).to(BF16).float()
    y.backward(gy.to(device).to(y.dtype))
    xr = x.clone().requires_grad_(True)
    gr, br = torch.tensor(g0, dtype=torch.float32, requires_grad=True), torch.tensor(b0, dtype=torch.float32, requires_grad=True)
    yr = torch.nn.functional.batch_norm(xr, None, None, gr, br, training=True, eps=2e-5)
    yr.backward(gy)
    def rel(a, b):
        return float((a - b).abs().max() / (b.abs().max() + 1e-12))
    assert rel(xd.grad.float().cpu(), xr.grad) < 3e-2
    assert rel(bn.gamma.grad.cpu(), gr.grad) < 2e-2 and rel(bn.beta.grad.cpu(), br.grad) < 2e-2
    # test mode uses the running statistics
    F.train_mode[0] = False
    try:
        with torch.no_grad():
            yt = bn(x.to(device))
    finally:
        F.train_mode[0] = True
    want_t, _, _ = onn.batch_normalization(x.numpy().astype(np.float64), g0, b0, am, av, train=False)
    np.testing.assert_allclose(yt.float().cpu().numpy(), want_t, rtol=2e-2, atol=2e-2)
    # 3-d input (B, C, T)
    bn3 = nn.BatchNormalization(8).to_gpu(0)
    x3 = torch.from_numpy(rs.randn(4, 8, 11).astype(np.float32)).to(BF16).float()
    y3 = bn3(x3.to(device))
    w3, _, _ = onn.batch_normalization(x3.numpy().astype(np.float64), np.ones(8), np.zeros(8), np.zeros(8), np.ones(8))
    np.testing.assert_allclose(y3.float().detach().cpu().numpy(), w3, rtol=2e-2, atol=2e-2)


def test_bf16_gradient_handover_from_layernorm_to_projection(device):
    """the float32 logit projection takes the LayerNormalization gradient in bf16 through its mailbox (no float32 dx, no
    cast pass); the gradients must equal those of the plain float32 route, also when the projection's output has a
    second consumer (autograd's part + the mailbox's part)"""
    from asr import functions as F, nn
    torch.manual_seed(11)
    B, C, T, V = 3, 32, 9, 24
    x = torch.randn(B, C, 1, T).to(device)
    gy = torch.randn(B, V, 1, T).to(device)

    def run(handover, second_consumer):
        torch.manual_seed(5)
        proj = nn.Convolution2D(C, V, (1, 1)).to_gpu()
        norm = nn.LayerNormalization(V).to_gpu()
        proj.output_float32 = True
        norm.output_float32 = True
        saved = F._producer_mailbox
        posted = []
        if not handover:
            F._producer_mailbox = lambda x2: None
        else:
            def spy(x2):
                box = saved(x2)
                posted.append(box)
                return box
            F._producer_mailbox = spy
        try:
            xin = x.clone().requires_grad_(True)
            h = proj(xin)
            assert h.dtype == torch.float32
            y = norm(h)
            loss = (y * gy).sum()
            if second_consumer:
                loss = loss + (h * 0.5).sum()
            loss.backward()
        finally:
            F._producer_mailbox = saved
        if handover:
            assert len(posted) == 1 and posted[0] is not None and posted[0].value is None     # found, used and emptied
        return [xin.grad.float().cpu()] + [p.grad.float().cpu().clone() for p in list(proj.parameters()) + list(norm.parameters())]

    for second in (False, True):
        a, b = run(True, second), run(False, second)
        for u, v in zip(a, b):
            assert u.shape == v.shape
            # both routes round dx to bf16 once (the f32 and bf16 kernels may differ in the last f32 bit before that);
            # with a second consumer the sum is rounded once more
            scale = float(v.abs().max()) + 1e-6
            torch.testing.assert_close(u, v, rtol=2e-2, atol=(2e-2 if second else 4e-3) * scale)


def test_layernorm_and_weightnorm_against_reference_goldens(device, golden_dir):
    """HIP layer-norm forward (f32 in, f32 out) and weight-norm kernel against the REFERENCE's outputs
    (tests/golden/norm.npz: NormalizeLayer.forward asr/nn/layernorm.py:33-48, _norm asr/nn/convolution_2d.py:21-25,62-64)."""
    import os
    import asr.nn as nn
    from asr import _ops
    g = np.load(os.path.join(golden_dir, "norm.npz"))
    for name in ("x4", "x3", "x4_wide", "x4_f64"):
        x = torch.tensor(g[name].astype(np.float32))
        ln = nn.LayerNormalization(x.shape[1]).to_gpu()          # gamma = 1, beta = 0: the bare normalize_layer
        ln.output_float32 = True
        # float32 input in physical order so that nothing is rounded to bf16 on the way in
        if x.dim() == 4:
            xd = x.to(device).permute(3, 0, 2, 1).contiguous().permute(1, 3, 2, 0)
        else:
            xd = x.to(device).permute(0, 2, 1).contiguous().permute(0, 2, 1)
        y = ln(xd)
        assert y.dtype == torch.float32 and tuple(y.shape) == tuple(x.shape)
        np.testing.assert_allclose(y.detach().cpu().numpy(), g[name + ".y"], rtol=2e-5, atol=2e-5)
        from asr.nn.layernorm import normalize_layer
        y2 = normalize_layer(xd)
        np.testing.assert_allclose(y2.detach().float().cpu().numpy(), g[name + ".y"], rtol=1e-2, atol=1e-2)   # bf16 output
    W, norm = _ops.weightnorm_fwd(torch.tensor(g["V"]).to(device), torch.tensor(g["g"]).to(device))
    np.testing.assert_allclose(norm.cpu().numpy(), g["norm"].reshape(-1), rtol=1e-6)
    np.testing.assert_allclose(W.cpu().numpy(), g["W"], rtol=2e-6, atol=1e-7)


def test_nstep_bigru_has_chainers_call_semantics(device):
    """hy, ys = NStepBiGRU(n_layers, in, out, dropout)(None, [x_0, x_1, ...]) with ragged (T_i, I) sequences: outputs
    (T_i, 2H) with the directions concatenated, hy (n_layers * 2, B, H); against torch.nn.GRU per sequence"""
    import asr.nn as nn
    torch.manual_seed(0)
    I, H, L = 16, 32, 2
    rnn = nn.NStepBiGRU(L, I, H, 0).to_gpu()
    lens = [9, 14, 9, 5]
    xs = [_bf(torch.randn(t, I)) for t in lens]
    hy, ys = rnn(None, [x.to(device) for x in xs])
    assert hy.shape == (L * 2, len(lens), H) and [tuple(y.shape) for y in ys] == [(t, 2 * H) for t in lens]
    ref = torch.nn.GRU(I, H, num_layers=L, bidirectional=True)
    with torch.no_grad():
        for layer in range(L):
            for d, suf in enumerate(("", "_reverse")):
                link = getattr(rnn, "l%d_%d" % (layer, d))
                getattr(ref, "weight_ih_l%d%s" % (layer, suf)).copy_(_bf(link.w_ih.detach().cpu()[0]))
                getattr(ref, "weight_hh_l%d%s" % (layer, suf)).copy_(_bf(link.w_hh.detach().cpu()[0]))
                getattr(ref, "bias_ih_l%d%s" % (layer, suf)).copy_(link.b_ih.detach().cpu()[0])
                getattr(ref, "bias_hh_l%d%s" % (layer, suf)).copy_(link.b_hh.detach().cpu()[0])
    for i, x in enumerate(xs):
        yr, hr = ref(x[:, None, :])
        assert _rel(ys[i].float().cpu(), yr[:, 0].detach()) < 2e-2, i
        assert _rel(hy[:, i].float().cpu(), hr[:, 0].detach()) < 2e-2, i
    loss = sum(y.float().sum() for y in ys)
    loss.backward()
    assert rnn.l0_1.w_hh.grad is not None and torch.isfinite(rnn.l1_0.w_ih.grad).all()
    with pytest.raises(ValueError):                 # (a given hx: tests/test_gru_state_gpu.py; a mis-shaped one is refused)
        rnn(torch.zeros(L * 2 + 1, len(lens), H, device=device), [x.to(device) for x in xs])


def test_projection_bias_gradient_with_a_second_consumer_of_the_logits(device):
    """the fused LayerNorm + CTC sweep adds the column sums of ITS dx to the projection's bias gradient and marks it
    done; a second consumer of the projection's output (a branch off the un-normalised logits) sends its gradient through autograd,
    and the projection still owes the bias the column sums of that part.  Checked against the route with every shortcut off."""
    from asr import functions as F, nn, _ops
    from asr.loss import connectionist_temporal_classification
    torch.manual_seed(3)
    B, C, T, V, L = 3, 32, 40, 24, 4
    x = torch.randn(B, C, 1, T).to(device)
    lab = torch.randint(1, V, (B, L), dtype=torch.int32).to(device)
    w = torch.randn(B, V, 1, T).to(device) * 0.05

    def run(shortcuts):
        torch.manual_seed(5)
        proj = nn.Convolution2D(C, V, (1, 1)).to_gpu()
        norm = nn.LayerNormalization(V).to_gpu()
        proj.output_float32 = norm.output_float32 = True
        keep = F.FUSE_CTC_INTO_LAYERNORM[0]
        F.FUSE_CTC_INTO_LAYERNORM[0] = shortcuts
        try:
            before = _ops.CALLS.get("layernorm_ctc_bwd", 0)
            h = proj(x)
            y = norm(h)
            tbv = y.permute(3, 0, 2, 1).squeeze(2)
            loss = connectionist_temporal_classification(tbv, lab, 0) + (h * w).sum()
            loss.backward()
            F.join_side_stream()
            torch.cuda.synchronize()
            assert (_ops.CALLS.get("layernorm_ctc_bwd", 0) - before) == (1 if shortcuts else 0)
            first = proj.b.grad.float().cpu().clone()
            # a second backward pass over a fresh graph starts afresh (the flags were reset when read)
            proj.cleargrads(); norm.cleargrads()
            h = proj(x)
            tbv = norm(h).permute(3, 0, 2, 1).squeeze(2)
            (connectionist_temporal_classification(tbv, lab, 0) + (h * w).sum()).backward()
            F.join_side_stream()
            torch.cuda.synchronize()
            assert torch.allclose(proj.b.grad.float().cpu(), first, rtol=1e-4, atol=1e-6)
            return first, proj.W.grad.float().cpu().clone()
        finally:
            F.FUSE_CTC_INTO_LAYERNORM[0] = keep
    (ba, wa), (bb, wb) = run(True), run(False)
    want_extra = w.float().sum(dim=(0, 2, 3)).cpu()           # d/db of (h * w).sum()
    assert float((bb - ba).norm()) <= 1e-2 * float(bb.norm()), (ba, bb)
    assert float(want_extra.norm()) > 10 * float((bb - ba).norm())        # the second consumer's share is what would be missing
    assert float((wa - wb).norm()) <= 2e-2 * float(wb.norm())


# ------------------------------------------------------------------------------------------------ SRU, chunked scans
@pytest.mark.parametrize("T,Bn,D,use_tanh,masked", [(77, 5, 96, True, False), (64, 3, 34, False, True), (200, 8, 128, True, True),
                                                    (33, 2, 64, True, False), (1000, 4, 64, False, False)])
def test_sru_chunked_scans_equal_the_column_scans(device, T, Bn, D, use_tanh, masked):
    """csrc/sru.hip: the time-chunked scans (summaries + apply) against the one-thread-per-column kernels of the reference's shape
    on the same inputs -- same formulas; the fast exp / rcp sigmoid and the chunk products change the last float32 bits only"""
    from asr import _ops, _lib
    from asr.nn.sru import sru
    from asr.link import Parameter
    assert _lib.lib().asr_sru_ws_bytes(T, Bn, D) > 0
    torch.manual_seed(T + D)
    X = _bf(torch.randn(Bn, D, T))
    W = _bf(torch.randn(3 * D, D) * (0.6 / np.sqrt(D)))
    Bias = torch.randn(2 * D) * 0.3
    c0 = torch.randn(Bn, D)
    mask = (torch.rand(Bn, D) > 0.3).float() if masked else None
    gH = _bf(torch.randn(Bn, D, T))
    gcT = torch.randn(Bn, D)
    res = []
    for chunked in (True, False):
        _ops.SRU_CHUNKED[0] = chunked
        try:
            Xd = X.to(device).requires_grad_(True)
            Wp, Bp = Parameter(W.to(device)), Parameter(Bias.to(device))
            c0d = c0.to(device).requires_grad_(True)
            H, C, cT = sru(Xd, Wp, Bp, c0d, use_tanh, None if mask is None else mask.to(device))
            ((H.float() * gH.to(device)).sum() + (cT * gcT.to(device)).sum()).backward()
            torch.cuda.synchronize()
            res.append([t.detach().float().cpu() for t in (H, C, cT, Xd.grad, Wp.grad, Bp.grad, c0d.grad)])
        finally:
            _ops.SRU_CHUNKED[0] = True
    for name, a, b in zip(("H", "C", "cT", "gX", "gW", "gB", "gc0"), *res):
        tol = 1e-5 if name in ("C", "cT") else 4e-3          # bf16 outputs: a last-bit difference in float32 may flip a rounding
        assert _rel(a, b) < tol, (name, _rel(a, b))


@pytest.mark.parametrize("T,Bn,D,use_tanh", [(96, 4, 64, True), (40, 3, 128, False), (300, 2, 64, True)])
def test_sru_gradient_through_the_last_cell_state_only(device, T, Bn, D, use_tanh):
    """SRUFunction sets set_materialize_grads(False) (as the reference does, asr/nn/sru.py:372-376 treats a missing gH as zeros): a loss
    that reads only c_T hands the backward kernels gH = None.  The chunked scans then load a stand-in row (float32 cell states / x read as
    bf16 pairs) whose bit patterns include Inf and NaN -- the value must be SELECTED to zero, not scaled by it (0 * Inf = NaN spreads into
    every earlier chunk, gU, gc0 and the bias gradient; about 1 in 256 low halves of a float32 is such a pattern, so every shape here has
    dozens).  chunked == column scans == float64 oracle with gH = 0."""
    from asr import _ops
    from asr.nn.sru import sru
    from asr.link import Parameter
    torch.manual_seed(3 * T + D)
    X = _bf(torch.randn(Bn, D, T))
    W = _bf(torch.randn(3 * D, D) * (0.6 / np.sqrt(D)))
    Bias = torch.randn(2 * D) * 0.3
    c0 = torch.randn(Bn, D)
    gcT = torch.randn(Bn, D)
    res = []
    for chunked in (True, False):
        _ops.SRU_CHUNKED[0] = chunked
        try:
            Xd = X.to(device).requires_grad_(True)
            Wp, Bp = Parameter(W.to(device)), Parameter(Bias.to(device))
            c0d = c0.to(device).requires_grad_(True)
            H, C, cT = sru(Xd, Wp, Bp, c0d, use_tanh)
            (cT * gcT.to(device)).sum().backward()
            torch.cuda.synchronize()
            res.append([t.detach().float().cpu() for t in (Xd.grad, Wp.grad, Bp.grad, c0d.grad)])
        finally:
            _ops.SRU_CHUNKED[0] = True
    for name, a, b in zip(("gX", "gW", "gB", "gc0"), *res):
        assert bool(torch.isfinite(a).all()) and bool(torch.isfinite(b).all()), name
        assert _rel(a, b) < 4e-3, (name, _rel(a, b))
    args = (X.double().numpy(), W.double().numpy(), Bias.double().numpy(), c0.double().numpy())
    gX, gW, gb, gc = onn.sru_bwd(*args, np.zeros((Bn, D, T)), gcT.double().numpy(), use_tanh)
    for name, a, r in zip(("gX", "gW", "gB", "gc0"), res[0], (gX, gW, gb, gc)):
        assert _rel(a, r) < 1e-2, (name, _rel(a, r))


@pytest.mark.parametrize("use_tanh", [True, False])
def test_sru_full_size_against_the_oracle(device, use_tanh):
    """T=1000, B=32, D=512 (the size tools/time_sru.py measures): H, C, c_T and every gradient against oracle.nn.sru_fwd / sru_bwd
    (asr/nn/sru.py:289-324 forward_cpu restated, pinned by tests/golden/sru.npz; backward = kernel K2 :75-191 restated)"""
    from asr.nn.sru import sru
    from asr.link import Parameter
    torch.manual_seed(7)
    Bn, D, T = 32, 512, 1000
    X = _bf(torch.randn(Bn, D, T))
    W = _bf(torch.randn(3 * D, D) * (0.6 / np.sqrt(D)))
    Bias = torch.randn(2 * D) * 0.3
    c0 = torch.randn(Bn, D)
    gH = _bf(torch.randn(Bn, D, T))
    gcT = torch.randn(Bn, D)
    Xd = X.to(device).requires_grad_(True)
    Wp, Bp = Parameter(W.to(device)), Parameter(Bias.to(device))
    c0d = c0.to(device).requires_grad_(True)
    H, C, cT = sru(Xd, Wp, Bp, c0d, use_tanh)
    ((H.float() * gH.to(device)).sum() + (cT * gcT.to(device)).sum()).backward()
    torch.cuda.synchronize()
    args = (X.double().numpy(), W.double().numpy(), Bias.double().numpy(), c0.double().numpy())
    Hr, Cr, cTr = onn.sru_fwd(*args, use_tanh)
    assert _rel(C.cpu(), Cr) < 2e-5 and _rel(cT.cpu(), cTr) < 2e-5
    assert _rel(H.float().cpu(), Hr) < 5e-3
    gX, gW, gb, gc = onn.sru_bwd(*args, gH.double().numpy(), gcT.double().numpy(), use_tanh)
    errs = dict(gX=_rel(Xd.grad.float().cpu(), gX), gW=_rel(Wp.grad.cpu(), gW), gB=_rel(Bp.grad.cpu(), gb), gc0=_rel(c0d.grad.cpu(), gc))
    print("full-size SRU (tanh=%s) vs float64 oracle:" % use_tanh, {k: "%.2e" % v for k, v in errs.items()})
    for k, v in errs.items():
        assert v < 1e-2, (k, v)


# ------------------------------------------------------------------------------------------------ the rest of asr.nn's function layers
def test_crelu_softmax_pooling_layers(device):
    """asr/nn/nn.py:18-23 CReLU, :42-43 LogSoftmax, :58-63 Softmax, :77-93 AveragePooling2D / ND, :105-113 MaxPoolingND, :123-133
    Unpooling2D against their torch-CPU statements (Chainer functions in the reference: unpinned boundary), forward and backward"""
    import asr.nn as nn
    F = torch.nn.functional
    x = _img(2, 6, 7, 5) * 2
    gy_seed = 11

    def both(layer, ref, out_channels=None):
        xd = x.to(device).requires_grad_(True)
        y = layer(xd)
        xr = x.clone().requires_grad_(True)
        yr = ref(xr)
        assert tuple(y.shape) == tuple(yr.shape), (y.shape, yr.shape)
        gy = _bf(torch.randn(yr.shape, generator=torch.Generator().manual_seed(gy_seed)))
        y.float().backward(gy.to(device))
        yr.backward(gy)
        return _rel(y.float().cpu(), yr.detach()), _rel(xd.grad.float().cpu(), xr.grad)

    e = both(nn.CReLU(), lambda t: torch.cat([F.relu(t), F.relu(-t)], dim=1))
    assert e[0] == 0.0 and e[1] < 1e-6, e
    e = both(nn.Softmax(), lambda t: torch.softmax(t, dim=1))
    assert e[0] < 4e-3 and e[1] < 1e-2, e
    e = both(nn.LogSoftmax(), lambda t: torch.log_softmax(t, dim=1))
    assert e[0] < 4e-3 and e[1] < 1e-2, e
    for k in (2, 3, 7):
        e = both(nn.AveragePooling2D((k, 1)), lambda t: F.avg_pool2d(t, (k, 1), stride=(k, 1)))
        assert e[0] < 4e-3 and e[1] < 4e-3, (k, e)
    e = both(nn.AveragePoolingND((2, 1)), lambda t: F.avg_pool2d(t, (2, 1), stride=(2, 1)))
    assert e[0] < 4e-3 and e[1] < 4e-3, e
    e = both(nn.MaxPoolingND((3, 1)), lambda t: F.max_pool2d(t, (3, 1), stride=(3, 1), ceil_mode=True))
    assert e[0] == 0.0 and e[1] < 1e-6, e
    for cover_all, Hout in ((True, 3 * 6 + 1), (False, 3 * 7)):
        e = both(nn.Unpooling2D((3, 1), cover_all=cover_all), lambda t: torch.repeat_interleave(t, 3, dim=2)[:, :, :Hout])
        assert e[0] == 0.0 and e[1] < 4e-3, (cover_all, e)
    with pytest.raises(NotImplementedError):
        nn.Softmax(axis=2)(x.to(device))
    with pytest.raises(NotImplementedError):          # (pooling over time is outside the height-pooling layers; UpSampling2D and
        nn.UpSampling2D(None, (2, 2))(x.to(device))   # SpatialPyramidPooling2D themselves: test_upsampling_and_spatial_pyramid_pooling)
    # 3-d (B, D, T) and 2-d inputs take the same kernels
    x3 = _bf(torch.randn(3, 10, 8))
    y3 = nn.Softmax()(x3.to(device))
    assert _rel(y3.float().cpu(), torch.softmax(x3, dim=1)) < 4e-3


def test_gaussian_noise_layer(device):
    """asr/nn/nn.py:220-231: x + N(0, std^2) when training (the `mean` argument is unused there too), identity otherwise;
    the gradient passes through unchanged"""
    import asr.nn as nn
    from asr import functions as F
    torch.manual_seed(0)
    x = torch.zeros(4, 64, 8, 128)
    layer = nn.GaussianNoise(0.0, 0.25)
    xd = x.to(device).requires_grad_(True)
    y = layer(xd)
    y.float().sum().backward()
    d = y.float().cpu()
    assert abs(float(d.mean())) < 2e-3 and abs(float(d.std()) - 0.25) < 3e-3
    kurt = float(((d / d.std()) ** 4).mean())
    assert abs(kurt - 3.0) < 0.05, kurt                      # Gaussian, not uniform
    assert torch.equal(xd.grad.cpu(), torch.ones_like(x))
    y2 = layer(xd)
    assert not torch.equal(y2, y)                            # a fresh draw per call
    F.train_mode[0] = False
    try:
        assert layer(xd) is xd
    finally:
        F.train_mode[0] = True


def test_upsampling_and_spatial_pyramid_pooling(device):
    """asr/nn/nn.py:135-146 UpSampling2D (chainer.functions.upsampling_2d: the inverse of a max pooling given its argmax `indexes`) and
    :115-121 SpatialPyramidPooling2D (max pooling over 2^l x 2^l bins of the (H, T) plane per level) -- Chainer functions in the reference
    (unpinned boundary): checked against their definitions in numpy / torch-CPU, forward and backward"""
    import asr.nn as nn
    from asr import functions as F
    B, C, H, T, k = 2, 8, 7, 5, 3
    x = _img(B, C, H, T, 3)
    xd = x.to(device).to(BF16)
    # indexes of max_pooling_2d(x, (k, 1)) (cover_all: Hout = 3) and the pooled values
    idx = F.max_pooling_2d_indexes(xd, (k, 1))
    pooled = F.max_pooling_2d(xd, (k, 1))
    Hp = pooled.shape[2]
    xp = torch.full((B, C, Hp * k, T), float("-inf"))
    xp[:, :, :H] = x
    win = xp.reshape(B, C, Hp, k, T)
    assert torch.equal(idx.cpu().long(), win.argmax(dim=3))
    # upsampling the pooled tensor with those indexes puts every maximum back where it came from, zeros elsewhere
    layer = nn.UpSampling2D(idx, (k, 1), outsize=(H, T), cover_all=True)
    pd = pooled.detach().clone().requires_grad_(True)
    up = layer(pd)
    assert tuple(up.shape) == (B, C, H, T)
    want = torch.zeros(B, C, Hp * k, T)
    want.scatter_(2, (torch.arange(Hp).reshape(1, 1, Hp, 1) * k + idx.cpu().long()), pooled.detach().float().cpu())
    assert torch.equal(up.detach().float().cpu(), want[:, :, :H])
    gy = _img(B, C, H, T, 4)
    up.backward(gy.to(device).to(BF16))
    gyp = torch.zeros(B, C, Hp * k, T)
    gyp[:, :, :H] = gy
    assert torch.equal(pd.grad.float().cpu(), gyp.gather(2, torch.arange(Hp).reshape(1, 1, Hp, 1) * k + idx.cpu().long()))
    # spatial pyramid pooling, three levels, on a plane whose sizes are not multiples of the bin counts
    B, C, H, T, height = 2, 8, 13, 37, 3
    x = _img(B, C, H, T, 5)
    xr = x.clone().requires_grad_(True)
    outs = []
    for l in range(height):
        nb = 2 ** l
        kh, kw = -(-H // nb), -(-T // nb)
        ph, pw = (nb * kh - H + 1) // 2, (nb * kw - T + 1) // 2
        xpad = torch.nn.functional.pad(xr, (pw, nb * kw - T - pw, ph, nb * kh - H - ph), value=float("-inf"))
        outs.append(xpad.reshape(B, C, nb, kh, nb, kw).amax(dim=(3, 5)).reshape(B, C * nb * nb))
    yr = torch.cat(outs, dim=1).reshape(B, -1, 1, 1)
    xd = x.to(device).to(BF16).requires_grad_(True)
    y = nn.SpatialPyramidPooling2D(height, nn.MaxPooling2D)(xd)
    assert tuple(y.shape) == tuple(yr.shape) == (B, C * 21, 1, 1)
    assert torch.equal(y.detach().float().cpu(), yr.detach())
    g = _bf(torch.randn(yr.shape, generator=torch.Generator().manual_seed(8)))
    y.backward(g.to(device).to(BF16))
    # the gradient of a bin goes to its FIRST maximum in (h, t) order (Chainer's argmax over the flattened window; bf16 inputs tie often,
    # and torch's amax would split the gradient between equal values)
    want, o = torch.zeros(B, C, H, T), 0
    gflat = g.reshape(B, -1)
    for l in range(height):
        nb = 2 ** l
        kh, kw = -(-H // nb), -(-T // nb)
        ph, pw = (nb * kh - H + 1) // 2, (nb * kw - T + 1) // 2
        gl = gflat[:, o:o + C * nb * nb].reshape(B, C, nb, nb)
        o += C * nb * nb
        for by in range(nb):
            for bx in range(nb):
                h0, h1 = max(0, by * kh - ph), min(H, by * kh - ph + kh)
                t0, t1 = max(0, bx * kw - pw), min(T, bx * kw - pw + kw)
                w = x[:, :, h0:h1, t0:t1].reshape(B, C, -1)
                a = w.argmax(dim=2)
                hh, tt = h0 + a // (t1 - t0), t0 + a % (t1 - t0)
                for b in range(B):
                    want[b, torch.arange(C), hh[b], tt[b]] += gl[b, :, by, bx]
    assert _rel(xd.grad.float().cpu(), want) < 6e-3              # (up to three levels add into one element: one bf16 rounding of the sum)
