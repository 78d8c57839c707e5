"""A GRU layer with a given initial state (chainer.links.NStepGRU / NStepBiGRU: ``hy, ys = rnn(hx, xs)``, asr/nn/nn.py:3; the reference's
SRU model carries its contexts the same way, run/ctc/sru/model.py:105-122): asr_gru_fwd_state / asr_gru_bwd_state against the float32
statement of the layer with h0 (oracle/bf16.py: gru_f32_state = torch's CPU GRU), full and ragged, one and two directions -- outputs,
final state, and the gradients of the input, the initial state and every parameter -- plus the property the state exists for: a
sequence fed in pieces with the state carried equals the sequence fed whole (forward: bit for bit)."""
import numpy as np
import pytest
import torch

from oracle import bf16 as ob

pytestmark = pytest.mark.gpu
F32 = torch.float32


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def _link(ndir, I, H, seed):
    from asr import nn
    torch.manual_seed(seed)
    return (nn.GRU if ndir == 1 else nn.BiGRU)(I, H).to_gpu()


def _oracle_params(link):
    return [p.detach().float().cpu().clone().requires_grad_(True) for p in (link.w_ih, link.w_hh, link.b_ih, link.b_hh)]


@pytest.mark.parametrize("ndir,ragged", [(1, False), (2, False), (1, True), (2, True)])
def test_layer_with_initial_state_against_the_float32_oracle(device, ndir, ragged):
    from asr import _ops
    B, I, H, T = 5, 24, 64, 19
    link = _link(ndir, I, H, seed=3)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, I, T, generator=g)
    h0 = 0.6 * torch.randn(ndir, B, H, generator=g)
    wy, wh = torch.randn(B, H, T, generator=g), torch.randn(ndir, B, H, generator=g)
    x_len = torch.tensor([19, 11, 19, 4, 15], dtype=torch.int32) if ragged else None
    live = None if x_len is None else (torch.arange(T).reshape(1, 1, T) < x_len.reshape(B, 1, 1)).float()

    xd = x.to(device).to(_ops.BF16).requires_grad_(True)
    hd = h0.to(device).requires_grad_(True)
    y, hy = link(xd, x_length=None if x_len is None else x_len.to(device), hx=hd)
    assert y.shape == (B, H, T) and hy.shape == (ndir, B, H) and hy.dtype == F32
    loss = (y.float() * wy.to(device)).sum() + (hy * wh.to(device)).sum()
    loss.backward()
    from asr.functions import join_side_stream
    join_side_stream()
    torch.cuda.synchronize()

    w_ih, w_hh, b_ih, b_hh = _oracle_params(link)
    xr = ob.rnd(x).permute(2, 0, 1).contiguous().requires_grad_(True)          # (T, B, I): the device sees the input in 16 bits
    hr = h0.clone().requires_grad_(True)
    yr, hyr = ob.gru_f32_state(xr, w_ih, w_hh, b_ih, b_hh, hr, x_len)
    lr = (yr.permute(1, 2, 0) * wy * (1.0 if live is None else live)).sum() + (hyr * wh).sum()
    lr.backward()
    got_y = y.detach().float().cpu() * (1.0 if live is None else live)
    errs = {"y": _rel(got_y, yr.detach().permute(1, 2, 0) * (1.0 if live is None else live)), "hy": _rel(hy.detach().cpu(), hyr.detach()),
            "dx": _rel(xd.grad.float().cpu() * (1.0 if live is None else live), xr.grad.permute(1, 2, 0)),
            "dhx": _rel(hd.grad.cpu(), hr.grad)}
    for name, p, r in (("w_ih", link.w_ih, w_ih), ("w_hh", link.w_hh, w_hh), ("b_ih", link.b_ih, b_ih), ("b_hh", link.b_hh, b_hh)):
        errs["d" + name] = _rel(p.grad.float().cpu(), r.grad)
    print("GRU with initial state, ndir=%d ragged=%s:" % (ndir, ragged), ", ".join("%s %.1e" % kv for kv in errs.items()))
    # bf16 state operand of the recurrent product, bf16 outputs and gate gradients against an all-float32 layer
    # (2 x measured: y 2.1e-3, hy 1.5e-3, gradients 1.9e-3 .. 3.4e-3)
    assert errs["y"] < 4.5e-3 and errs["hy"] < 3.5e-3, errs
    for k in ("dx", "dhx", "dw_ih", "dw_hh", "db_ih", "db_hh"):
        assert errs[k] < 7e-3, (k, errs)


def test_sequence_fed_in_pieces_equals_the_sequence_fed_whole(device):
    """unidirectional layer, T = 23 as 9 + 14 frames with the state carried: outputs, final state AND gradients (the gradient of the second
    piece's initial state is what flows into the first piece's final state) equal those of one call -- the forward pass bit for bit (same
    kernels, same order of operations; the carried state is the float32 one, rounded for the product exactly as inside a call), the
    gradients to float32 rounding"""
    from asr import _ops
    B, I, H, T, T1 = 4, 16, 64, 23, 9
    link = _link(1, I, H, seed=5)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, I, T, generator=g).to(device).to(_ops.BF16)
    h0 = (0.5 * torch.randn(1, B, H, generator=g)).to(device)
    wy = torch.randn(B, H, T, generator=g).to(device)

    def run(pieces):
        for p in link.parameters():
            p.grad = None
        xin = x.clone().requires_grad_(True)
        hin = h0.clone().requires_grad_(True)
        ys, h, t0 = [], hin, 0
        for n in pieces:
            y, h = link(xin[:, :, t0:t0 + n].contiguous(), hx=h)
            ys.append(y)
            t0 += n
        y = torch.cat(ys, dim=2)
        ((y.float() * wy).sum() + h.sum()).backward()
        from asr.functions import join_side_stream
        join_side_stream()
        torch.cuda.synchronize()
        return [t.detach().clone() for t in (y, h, xin.grad, hin.grad, link.w_ih.grad, link.b_hh.grad)], link.w_hh.grad.detach().clone()

    whole, gwhh_whole = run([T])
    parts, gwhh_parts = run([T1, T - T1])
    for name, a, b in zip(("y", "hy", "dx", "dhx", "dw_ih", "db_hh"), whole, parts):
        if name in ("dw_ih", "db_hh", "dx", "dhx"):
            # weight gradients: sums over time of the same per-step terms, formed in two products instead of one; state / input gradients:
            # at the seam dh_{t-1} = dh_t z + dgh_t W_hh is formed by asr_gru_bwd_state's closing kernel (a float32 dot product) instead
            # of the step kernel's MFMA partial sums -- the last float32 bit
            assert _rel(b, a) < 1e-5, name
        else:
            assert torch.equal(a, b), (name, float((a.float() - b.float()).abs().max()))
    assert _rel(gwhh_parts, gwhh_whole) < 1e-5


def test_zero_initial_state_equals_the_default_kernels(device):
    """hx = 0 through the per-step kernels against the layer called without a state (the persistent kernels): same layer, different
    kernels and a bf16 instead of a float32 input projection -- bf16-level agreement"""
    from asr import _ops
    B, I, H, T = 6, 32, 128, 40
    link = _link(2, I, H, seed=9)
    x = torch.randn(B, I, T, generator=torch.Generator().manual_seed(2)).to(device).to(_ops.BF16)
    y0 = link(x)
    y1, hy = link(x, hx=torch.zeros(2, B, H, device=device))
    assert _rel(y1.float(), y0.float()) < 1e-2
    assert hy.shape == (2, B, H) and bool(torch.isfinite(hy).all())


def test_nstep_gru_with_initial_state(device):
    """chainer's call: hy, ys = rnn(hx, xs) with ragged xs and a given hx, against torch.nn.GRU with the same parameters"""
    from asr import nn
    torch.manual_seed(4)
    n_layers, I, H = 2, 12, 32
    rnn = nn.NStepBiGRU(n_layers, I, H, 0.0).to_gpu()
    lens = [9, 5, 9, 7]
    g = torch.Generator().manual_seed(3)
    xs = [torch.randn(n, I, generator=g) for n in lens]
    hx = 0.5 * torch.randn(n_layers * 2, len(xs), H, generator=g)
    hy, ys = rnn(hx.to(device), [x.to(device) for x in xs])
    assert hy.shape == (n_layers * 2, len(xs), H) and [tuple(y.shape) for y in ys] == [(n, 2 * H) for n in lens]
    ref = torch.nn.GRU(I, H, n_layers, bidirectional=True)
    with torch.no_grad():
        for layer in range(n_layers):
            for d in range(2):
                link = getattr(rnn, "l%d_%d" % (layer, d))
                sfx = "_l%d%s" % (layer, "_reverse" if d else "")
                getattr(ref, "weight_ih" + sfx).copy_(link.w_ih.detach().cpu()[0])
                getattr(ref, "weight_hh" + sfx).copy_(link.w_hh.detach().cpu()[0])
                getattr(ref, "bias_ih" + sfx).copy_(link.b_ih.detach().cpu()[0])
                getattr(ref, "bias_hh" + sfx).copy_(link.b_hh.detach().cpu()[0])
        packed = torch.nn.utils.rnn.pack_sequence([ob.rnd(x) for x in xs], enforce_sorted=False)
        out, hn = ref(packed, hx)
        out, _ = torch.nn.utils.rnn.pad_packed_sequence(out, batch_first=True)
    for i, n in enumerate(lens):
        assert _rel(ys[i].float().cpu(), out[i, :n]) < 2e-2, i
    assert _rel(hy.float().cpu(), hn) < 2e-2
