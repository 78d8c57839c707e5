"""HIP CTC / Gram-CTC through the C ABI against the CPU oracle and the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

from oracle import ctc as octc

pytestmark = pytest.mark.gpu

LOSS_RTOL = 1e-4            # BASELINE.json: "CTC loss matching reference to 1e-4 rel"
GRAD_ATOL, GRAD_RTOL = 1e-5, 1e-4


def _run(device, xs, uni, big, xl, tl, reduce, gy=None):
    from asr.loss import connectionist_temporal_classification, gram_ctc
    x = torch.tensor(xs, device=device, requires_grad=True)
    tu = torch.tensor(uni, device=device)
    txl = None if xl is None else torch.tensor(xl, device=device)
    ttl = None if tl is None else torch.tensor(tl, device=device)
    if big is None:
        loss = connectionist_temporal_classification(x, tu, 0, txl, ttl, reduce)
    else:
        loss = gram_ctc(x, tu, torch.tensor(big, device=device), 0, txl, ttl, reduce)
    if reduce == "mean":
        loss.backward(None if gy is None else torch.tensor(gy, device=device))
    else:
        loss.backward(torch.tensor(np.ones(xs.shape[1], np.float32) if gy is None else gy, device=device))
    return loss.detach().cpu().numpy(), x.grad.cpu().numpy()


NAMES = ["ctc_small", "ctc_noreduce", "ctc_full", "ctc_v300", "ctc_v3000", "ctc_len1",
         "gram_mixed", "gram_all", "gram_repeat2", "gram_v3000", "gram_len1"]


@pytest.mark.parametrize("name", NAMES)
def test_golden(device, golden_dir, name):
    g = np.load(os.path.join(golden_dir, "gram_ctc.npz"))
    xs, uni, big, xl, tl, loss, gy, gx = [g["%s.%s" % (name, k)] for k in ("xs", "uni", "big", "xl", "tl", "loss", "gy", "gx")]
    red = str(g[name + ".reduce"])
    l, gr = _run(device, xs, uni, big, xl, tl, red, gy)
    np.testing.assert_allclose(l, loss, rtol=LOSS_RTOL)
    np.testing.assert_allclose(gr, gx, rtol=2e-3, atol=2e-4)      # reference itself is float32 (see oracle test)
    # against the float64 oracle the fp32 tolerance of BASELINE.md applies
    lo, go = octc.gram_ctc_loss_grad(xs, uni, big, 0, xl, tl, red, gy)
    np.testing.assert_allclose(l, lo, rtol=LOSS_RTOL)
    np.testing.assert_allclose(gr, go, rtol=GRAD_RTOL, atol=GRAD_ATOL)
    if name.startswith("ctc"):
        l2, g2 = _run(device, xs, uni, None, xl, tl, red, gy)
        np.testing.assert_allclose(l2, lo, rtol=LOSS_RTOL)
        np.testing.assert_allclose(g2, go, rtol=GRAD_RTOL, atol=GRAD_ATOL)


@pytest.mark.parametrize("T,B,V,L,gram", [(50, 3, 7, 5, False), (200, 4, 119, 20, False), (400, 2, 3000, 40, False),
                                          (120, 3, 300, 15, True), (300, 2, 3001, 33, True), (64, 2, 8200, 9, False)])
def test_random_vs_oracle(device, T, B, V, L, gram):
    rs = np.random.RandomState(T + V)
    xs = (rs.randn(T, B, V) * 1.5).astype(np.float32)
    n_uni = min(V, 119)
    uni = rs.randint(1, n_uni, size=(B, L)).astype(np.int32)
    uni[:, 2] = uni[:, 1]
    tl = rs.randint(max(1, L // 2), L + 1, size=B).astype(np.int32)
    tl[0] = L
    xl = rs.randint(3 * L, T + 1, size=B).astype(np.int32)
    xl[0] = T
    big = None
    if gram:
        big = rs.randint(n_uni, V, size=(B, L)).astype(np.int32)
        big[rs.rand(B, L) < 0.3] = -1
        big[:, 0] = -1
    for red in ("mean", "no"):
        l, gr = _run(device, xs, uni, big, xl, tl, red)
        if gram:
            lo, go = octc.gram_ctc_loss_grad(xs, uni, big, 0, xl, tl, red)
        else:
            lo, go = octc.ctc_loss_grad(xs, uni, 0, xl, tl, red)
        np.testing.assert_allclose(l, lo, rtol=LOSS_RTOL)
        np.testing.assert_allclose(gr, go, rtol=GRAD_RTOL, atol=GRAD_ATOL)
        assert (gr[xl[1]:, 1] == 0).all()


def test_tuple_of_views_input(device):
    """The reference passes T separate (B, V) arrays; a tuple of views of one buffer is used without a copy."""
    from asr.loss import connectionist_temporal_classification
    rs = np.random.RandomState(1)
    T, B, V, L = 30, 3, 11, 4
    xs = rs.randn(T, B, V).astype(np.float32)
    lab = rs.randint(1, V, size=(B, L)).astype(np.int32)
    x = torch.tensor(xs, device=device, requires_grad=True)
    loss = connectionist_temporal_classification(tuple(x.unbind(0)), torch.tensor(lab, device=device), 0)
    loss.backward()
    lo, go = octc.ctc_loss_grad(xs, lab, 0)
    np.testing.assert_allclose(loss.item(), lo, rtol=LOSS_RTOL)
    np.testing.assert_allclose(x.grad.cpu().numpy(), go, rtol=GRAD_RTOL, atol=GRAD_ATOL)


def test_full_size_properties(device):
    """BASELINE size (T=1000, B=32, V=3000): properties that need no oracle run.
    rows of the gradient sum to 0 for t < x_len (softmax sums to 1 and so does the occupancy), are 0 beyond
    x_len, and the loss is invariant to adding a per-row constant to the logits."""
    from asr.loss import connectionist_temporal_classification
    T, B, V, L = 1000, 32, 3000, 120
    g = torch.Generator(device="cpu").manual_seed(0)
    xs = torch.randn(T, B, V, generator=g)
    lab = torch.randint(1, V, (B, L), generator=g, dtype=torch.int32)
    tl = torch.randint(40, L + 1, (B,), generator=g, dtype=torch.int32)
    xl = torch.randint(600, T + 1, (B,), generator=g, dtype=torch.int32)
    x = xs.to(device).requires_grad_(True)
    loss = connectionist_temporal_classification(x, lab.to(device), 0, xl.to(device), tl.to(device), "no")
    loss.sum().backward()
    gr = x.grad
    rowsum = gr.sum(dim=2).cpu()
    mask = (torch.arange(T)[:, None] < xl[None, :])
    assert rowsum[mask].abs().max().item() < 2e-4
    assert (gr.cpu()[~mask] == 0).all()
    shift = torch.randn(T, B, 1, generator=g).to(device)
    loss2 = connectionist_temporal_classification(x.detach() + shift, lab.to(device), 0, xl.to(device), tl.to(device), "no")
    np.testing.assert_allclose(loss2.cpu().numpy(), loss.detach().cpu().numpy(), rtol=1e-5)
    # one utterance of the full-size batch against the oracle
    b = 3
    lo, go = octc.ctc_loss_grad(xs[:, b:b + 1].numpy(), lab[b:b + 1].numpy(), 0, xl[b:b + 1].numpy(), tl[b:b + 1].numpy(), "no")
    np.testing.assert_allclose(loss[b].item(), lo[0], rtol=LOSS_RTOL)
    np.testing.assert_allclose(gr[:, b].cpu().numpy(), go[:, 0], rtol=GRAD_RTOL, atol=GRAD_ATOL)


# ---------------------------------------------------------------------------------------------- fused LayerNorm + CTC backward
def _ln_ctc_case(device, T, B, V, L, losses, seed, ragged=True):
    """x (T*B, V) -> per-frame LayerNormalization -> one or two CTC-family losses.  Returns what the graph gives for dx,
    dgamma, dbeta with the fused backward (recipes left at the normalisation) and with it switched off (the loss writes its
    (T, B, V) gradient, the normalisation reads it)."""
    from asr import functions as F
    from asr.link import Parameter
    from asr.loss import connectionist_temporal_classification, gram_ctc
    rs = np.random.RandomState(seed)
    x0 = torch.tensor((rs.randn(T * B, V) * 2.0 + 0.3).astype(np.float32)).to(device)
    g0 = torch.tensor(rs.uniform(0.5, 1.5, V).astype(np.float32))
    b0 = torch.tensor((rs.randn(V) * 0.2).astype(np.float32))
    n_uni = min(V, 40)
    uni = torch.tensor(rs.randint(1, n_uni, size=(B, L)).astype(np.int32)).to(device)
    big = rs.randint(n_uni, V, size=(B, L)).astype(np.int32) if V > n_uni else np.full((B, L), -1, np.int32)
    big[rs.rand(B, L) < 0.3] = -1
    big[:, 0] = -1
    big = torch.tensor(big).to(device)
    tl = torch.tensor((rs.randint(max(1, L // 2), L + 1, size=B) if ragged else np.full(B, L)).astype(np.int32)).to(device)
    xl = torch.tensor((rs.randint(max(3 * L + 2, T // 2), T + 1, size=B) if ragged else np.full(B, T)).astype(np.int32)).to(device)
    gy_no = torch.tensor(rs.rand(B).astype(np.float32)).to(device)

    def run(fused):
        F.FUSE_CTC_INTO_LAYERNORM[0] = fused
        try:
            x = x0.clone().requires_grad_(True)
            gamma, beta = Parameter(g0.clone().to(device)), Parameter(b0.clone().to(device))
            # logical (B, V, 1, T) view of physical (T, B, 1, V) float32 rows, as the models hand it to LayerNormalization
            xin = x.reshape(T, B, 1, V).permute(1, 3, 2, 0)
            y = F.layer_normalization(xin, gamma, beta, out_f32=True)
            tbv = y.permute(3, 0, 2, 1).squeeze(2)
            total = None
            for kind, reduce in losses:
                if kind == "ctc":
                    l = connectionist_temporal_classification(tbv, uni, 0, xl, tl, reduce)
                else:
                    l = gram_ctc(tbv, uni, big, 0, xl, tl, reduce)
                l = l if reduce == "mean" else (l * gy_no).sum()
                total = l if total is None else total + l
            total.backward()
            torch.cuda.synchronize()
            return total.item(), x.grad.clone(), gamma.grad.clone(), beta.grad.clone()
        finally:
            F.FUSE_CTC_INTO_LAYERNORM[0] = True

    return run(True), run(False)


@pytest.mark.parametrize("T,B,V,L,losses", [(40, 3, 28, 5, [("ctc", "mean")]), (60, 4, 120, 7, [("gram", "mean")]),
                                            (50, 2, 64, 6, [("ctc", "no")]), (70, 3, 200, 8, [("gram", "mean"), ("ctc", "mean")]),
                                            (300, 8, 3000, 40, [("ctc", "mean")]), (200, 4, 3000, 30, [("gram", "no"), ("ctc", "mean")])])
def test_layernorm_ctc_backward_fused_equals_unfused(device, T, B, V, L, losses):
    """csrc/ctc_ln.hip: the gradient with respect to the normalised logits formed inside the normalisation's backward sweep
    gives the dx / dgamma / dbeta of the two-kernel route (ctc::grad writes it, ln::bwd_rows_f32 reads it): same formulas,
    float32 throughout; the joint Gram-CTC + CTC case posts two recipes"""
    from asr import _ops
    before = _ops.CALLS.get("layernorm_ctc_bwd", 0)
    (lf, dxf, dgf, dbf), (lu, dxu, dgu, dbu) = _ln_ctc_case(device, T, B, V, L, losses, seed=T + V)
    assert _ops.CALLS.get("layernorm_ctc_bwd", 0) == before + 1          # the fused sweep ran exactly once (in the fused run)
    assert lf == lu
    scale = float(dxu.abs().max()) + 1e-30
    assert float((dxf - dxu).abs().max()) <= 2e-5 * scale + 1e-9, float((dxf - dxu).abs().max()) / scale
    for a, r in ((dgf, dgu), (dbf, dbu)):
        assert float((a - r).abs().max()) <= 1e-4 * (float(r.abs().max()) + 1e-30) + 1e-7


def test_layernorm_ctc_fusion_with_another_consumer_of_the_logits(device):
    """a second consumer of the normalised logits (here a plain torch expression) sends its gradient through autograd; the
    normalisation then adds the recipe's sweep and the ordinary backward sweep"""
    from asr import functions as F
    from asr.link import Parameter
    from asr.loss import connectionist_temporal_classification
    rs = np.random.RandomState(3)
    T, B, V, L = 30, 2, 36, 4
    x0 = torch.tensor(rs.randn(T * B, V).astype(np.float32)).to(device)
    uni = torch.tensor(rs.randint(1, V, size=(B, L)).astype(np.int32)).to(device)
    w = torch.tensor(rs.randn(T, B, V).astype(np.float32)).to(device)

    def run(fused):
        F.FUSE_CTC_INTO_LAYERNORM[0] = fused
        try:
            x = x0.clone().requires_grad_(True)
            gamma, beta = Parameter(torch.ones(V).to(device)), Parameter(torch.zeros(V).to(device))
            y = F.layer_normalization(x.reshape(T, B, 1, V).permute(1, 3, 2, 0), gamma, beta, out_f32=True)
            tbv = y.permute(3, 0, 2, 1).squeeze(2)
            total = connectionist_temporal_classification(tbv, uni, 0) + (tbv * w).sum() * 0.01
            total.backward()
            torch.cuda.synchronize()
            return x.grad.clone(), gamma.grad.clone()
        finally:
            F.FUSE_CTC_INTO_LAYERNORM[0] = True
    (dxf, dgf), (dxu, dgu) = run(True), run(False)
    assert float((dxf - dxu).abs().max()) <= 1e-4 * float(dxu.abs().max())
    assert float((dgf - dgu).abs().max()) <= 1e-4 * float(dgu.abs().max()) + 1e-6
