"""Oracle restatements (features, statistics, SRU, layer-norm) against the reference's golden vectors -- CPU only."""
import os

import numpy as np
import pytest

from oracle import fft as offt
from oracle import nn as onn


def test_filterbank_and_logmel_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "fft.npz"))
    fb = offt.get_filterbanks(40, 512, 16000)
    assert np.array_equal(fb, g["fbank"])
    assert fb.sum() == pytest.approx(247.0) and np.count_nonzero(fb) == 454
    lm = offt.compute_logmel(g["pspec"], fb)
    np.testing.assert_allclose(lm, g["logmel_full"], rtol=0, atol=1e-12)
    assert np.allclose(lm[3], np.log(np.finfo(float).eps))                            # the all-zero frame
    a, d, dd = offt.compute_deltas(lm)
    np.testing.assert_allclose(a, g["logmel"], atol=1e-12)
    np.testing.assert_allclose(d, g["delta"], atol=1e-12)
    np.testing.assert_allclose(dd, g["delta_delta"], atol=1e-12)
    assert offt.hz2mel(1000.0) == pytest.approx(float(g["hz2mel_1000"])) and offt.mel2hz(1000.0) == pytest.approx(float(g["mel2hz_1000"]))
    big = np.random.RandomState(0).rand(1002, 257)
    lb, db, ddb = offt.compute_deltas(offt.compute_logmel(big, fb))
    np.testing.assert_allclose([lb.mean(), db.std(), ddb.std()], g["big_stats"], rtol=1e-12)


def test_specgram_against_numpy_fft():
    """sigproc is absent (parity unpinned): framing + |rfft|^2/nfft checked against an independent numpy formulation."""
    rs = np.random.RandomState(1)
    sig = np.round(rs.randn(5000) * 3000).astype(np.int16)
    ps = offt.get_specgram(sig, 16000, 0.032, 0.01, 512, 0.97, np.hanning)
    assert ps.shape == (1 + int(np.ceil((5000 - 512) / 160)), 257)
    pre = np.concatenate([[float(sig[0])], sig[1:].astype(np.float64) - 0.97 * sig[:-1].astype(np.float64)])
    f = 7
    frame = pre[f * 160:f * 160 + 512] * np.hanning(512)
    np.testing.assert_allclose(ps[f], np.abs(np.fft.fft(frame)[:257]) ** 2 / 512, rtol=1e-10)
    last = np.zeros(512)
    tail = pre[(ps.shape[0] - 1) * 160:]
    last[:len(tail)] = tail
    np.testing.assert_allclose(ps[-1], np.abs(np.fft.rfft(last * np.hanning(512))) ** 2 / 512, rtol=1e-9, atol=1e-6)


def test_running_stats_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "stats.npz"))
    st = offt.RunningStats()
    for i in range(4):
        st.update(g["chunk%d" % i])
    np.testing.assert_allclose(st.mean, g["mean"], rtol=1e-6)
    np.testing.assert_allclose(st.nvar, g["nvar"], rtol=1e-5)
    assert st.total == int(g["total"])
    m, s = st.mean_and_std()
    np.testing.assert_allclose(m, g["bmean"], rtol=1e-6)
    np.testing.assert_allclose(s, g["bstd"], rtol=1e-5)
    allx = np.concatenate([g["chunk%d" % i] for i in range(4)], axis=2).astype(np.float64)
    np.testing.assert_allclose(st.mean, allx.mean(axis=2), atol=1e-5)
    np.testing.assert_allclose(s[0, ..., 0], allx.std(axis=2, ddof=1), rtol=1e-4)


def test_augmentation_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "augment.npz"))
    np.random.seed(int(g["seed_both"]))
    assert np.array_equal(offt.augment_specgram(g["pspec"].copy(), True, True), g["aug_both"])
    np.random.seed(int(g["seed_speed"]))
    assert np.array_equal(offt.augment_specgram(g["pspec"].copy(), True, False), g["aug_speed"])


@pytest.mark.parametrize("name", ["tanh", "linear", "masked"])
def test_sru_forward_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, "sru.npz"))
    X, W, B, c0, mask = (g["%s.%s" % (name, k)] for k in ("X", "W", "B", "c0", "mask"))
    H, C, cT = onn.sru_fwd(X.astype(np.float64), W.astype(np.float64), B.astype(np.float64), c0.astype(np.float64),
                           bool(g[name + ".use_tanh"]), mask.astype(np.float64))
    np.testing.assert_allclose(H, g[name + ".H"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(C, g[name + ".C"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(cT, g[name + ".cT"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("use_tanh", [True, False])
def test_sru_backward_by_finite_differences(use_tanh):
    """the reference has no CPU backward (asr/nn/sru.py:369-370): the restated kernel K2 is pinned to the pinned forward
    by central differences, the method of the reference's own test (asr/nn/test_layernorm.py:70-74)."""
    rs = np.random.RandomState(3)
    Bn, D, T = 2, 4, 5
    X, W = rs.randn(Bn, D, T), rs.randn(3 * D, D) * 0.5
    Bias, c0 = rs.randn(2 * D) * 0.3, rs.randn(Bn, D)
    gH, gcT = rs.randn(Bn, D, T), rs.randn(Bn, D)

    def loss(X_, W_, B_, c_):
        H, C, cT = onn.sru_fwd(X_, W_, B_, c_, use_tanh)
        return (H * gH).sum() + (cT * gcT).sum()
    gX, gW, gb, gc = onn.sru_bwd(X, W, Bias, c0, gH, gcT, use_tanh)
    eps = 1e-6
    for arr, grad, idx in ((X, gX, (1, 2, 3)), (W, gW, (5, 1)), (Bias, gb, (6,)), (c0, gc, (0, 3))):
        a = arr.copy()
        a[idx] += eps
        up = loss(*(a if arr is q else q for q in (X, W, Bias, c0)))
        a[idx] -= 2 * eps
        dn = loss(*(a if arr is q else q for q in (X, W, Bias, c0)))
        assert (up - dn) / (2 * eps) == pytest.approx(grad[idx], rel=1e-5, abs=1e-7)


def test_layernorm_backward_by_finite_differences():
    """asr/nn/test_layernorm.py:70-74: numerical gradient check of NormalizeLayer in float64."""
    rs = np.random.RandomState(0)
    x = rs.uniform(-10, 10, (2, 3, 4, 5))
    gy = rs.uniform(-1, 1, x.shape)
    y, diff, std = onn.normalize_layer_fwd(x)
    assert np.allclose(y.mean(axis=(1, 2)), 0) and np.allclose(y.std(axis=(1, 2)), 1)
    gx = onn.normalize_layer_bwd(gy, diff, std)
    eps = 1e-6
    for idx in ((0, 1, 2, 3), (1, 0, 0, 0), (1, 2, 3, 4)):
        xp, xm = x.copy(), x.copy()
        xp[idx] += eps
        xm[idx] -= eps
        num = ((onn.normalize_layer_fwd(xp)[0] - onn.normalize_layer_fwd(xm)[0]) * gy).sum() / (2 * eps)
        assert num == pytest.approx(gx[idx], rel=1e-5, abs=1e-8)


def test_weightnorm_backward_by_finite_differences():
    rs = np.random.RandomState(2)
    V, g = rs.randn(4, 3, 2, 2), rs.rand(4, 1, 1, 1) + 0.5
    gW = rs.randn(*V.shape)
    gV, gg = onn.weightnorm_bwd(gW, V, g)
    eps = 1e-6
    Vp, Vm = V.copy(), V.copy()
    Vp[1, 2, 0, 1] += eps
    Vm[1, 2, 0, 1] -= eps
    num = ((onn.weightnorm_W(Vp, g)[0] - onn.weightnorm_W(Vm, g)[0]) * gW).sum() / (2 * eps)
    assert num == pytest.approx(gV[1, 2, 0, 1], rel=1e-5)
    gp, gm = g.copy(), g.copy()
    gp[2] += eps
    gm[2] -= eps
    num = ((onn.weightnorm_W(V, gp)[0] - onn.weightnorm_W(V, gm)[0]) * gW).sum() / (2 * eps)
    assert num == pytest.approx(gg[2, 0, 0, 0], rel=1e-5)


def test_normalize_layer_and_weightnorm_match_reference(golden_dir):
    """oracle.nn against the reference's own NumPy statements (tests/golden/norm.npz, written by make_golden.py G9):
    NormalizeLayer.forward asr/nn/layernorm.py:33-48 (4-d, 3-d, f32/f64; no epsilon) and _norm / W = g V / _norm(V)
    asr/nn/convolution_2d.py:21-25,62-64"""
    g = np.load(os.path.join(golden_dir, "norm.npz"))
    for name in ("x4", "x4_f64", "x3", "x4_wide"):
        x = g[name]
        y, diff, std = onn.normalize_layer_fwd(x.astype(np.float64))
        tol = 1e-12 if x.dtype == np.float64 else 2e-6
        np.testing.assert_allclose(y, g[name + ".y"], rtol=tol, atol=tol * 5)
        np.testing.assert_allclose(std, g[name + ".std"], rtol=max(tol, 1e-6))
    W, Vn, norm = onn.weightnorm_W(g["V"].astype(np.float64), g["g"].astype(np.float64))
    np.testing.assert_allclose(norm, g["norm"], rtol=1e-6)
    np.testing.assert_allclose(W, g["W"], rtol=2e-6, atol=1e-7)
