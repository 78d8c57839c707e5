"""The CPU oracle against the reference's golden vectors (and torch CPU) -- no GPU needed."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc


def _cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "gram_ctc.npz"))
    return g, [str(n) for n in g["names"]]


def _get(g, n):
    return [g["%s.%s" % (n, k)] for k in ("xs", "uni", "big", "xl", "tl", "loss", "gy", "gx")] + [str(g[n + ".reduce"])]


@pytest.mark.parametrize("name", ["ctc_small", "ctc_noreduce", "ctc_full", "ctc_v300", "ctc_v3000", "ctc_len1",
                                  "gram_mixed", "gram_all", "gram_repeat2", "gram_v3000", "gram_len1"])
def test_gram_oracle_matches_reference(golden_dir, name):
    g, names = _cases(golden_dir)
    assert name in names
    xs, uni, big, xl, tl, loss, gy, gx, red = _get(g, name)
    l, gr = octc.gram_ctc_loss_grad(xs, uni, big, 0, xl, tl, red, gy)
    np.testing.assert_allclose(l, loss, rtol=1e-5)
    # the reference accumulates in float32 (error grows with T and V); the oracle is float64
    np.testing.assert_allclose(gr, gx, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("name", ["ctc_small", "ctc_noreduce", "ctc_full", "ctc_v300", "ctc_len1"])
def test_standard_ctc_is_gram_with_no_bigrams(golden_dir, name):
    g, _ = _cases(golden_dir)
    xs, uni, big, xl, tl, loss, gy, gx, red = _get(g, name)
    assert (big[:, :1] == -1).all()
    l, gr = octc.ctc_loss_grad(xs, uni, 0, xl, tl, red, gy)
    np.testing.assert_allclose(l, loss, rtol=1e-5)
    np.testing.assert_allclose(gr, gx, rtol=2e-3, atol=2e-4)


def test_standard_ctc_matches_torch_cpu():
    rs = np.random.RandomState(5)
    T, B, V, L = 40, 5, 17, 8
    xs = rs.randn(T, B, V).astype(np.float32) * 2
    lab = rs.randint(1, V, size=(B, L)).astype(np.int32)
    lab[:, 3] = lab[:, 2]
    tl = rs.randint(1, L + 1, size=B).astype(np.int32)
    xl = rs.randint(2 * L + 2, T + 1, size=B).astype(np.int32)
    loss, grad = octc.ctc_loss_grad(xs, lab, 0, xl, tl, "no")
    x = torch.tensor(xs, dtype=torch.float64, requires_grad=True)
    lt = torch.nn.functional.ctc_loss(torch.log_softmax(x, 2), torch.tensor(lab, dtype=torch.long), torch.tensor(xl, dtype=torch.long),
                                      torch.tensor(tl, dtype=torch.long), blank=0, reduction="none")
    lt.sum().backward()
    np.testing.assert_allclose(loss, lt.detach().numpy(), rtol=1e-9)
    np.testing.assert_allclose(grad, x.grad.numpy(), atol=1e-9)


def test_connection_matrices_match_reference(golden_dir):
    c = np.load(os.path.join(golden_dir, "gram_ctc_connection.npz"))
    for b in range(c["uni"].shape[0]):
        m = octc.gram_connection_matrix(c["uni"][b], c["big"][b], c["tl"][b], c["fwd"].shape[1])
        assert np.array_equal(m, c["fwd"][b])


def test_infeasible_alignment_is_flagged():
    xs = np.zeros((2, 1, 5), dtype=np.float32)          # T = 2 < 2 L + 1
    loss, grad = octc.ctc_loss_grad(xs, np.array([[1, 1, 2]], dtype=np.int32), 0, None, None, "no")
    assert loss[0] >= 1e9
