"""Each HIP kernel (through the C ABI) against the CPU oracle / torch-CPU fp32 on the same seeded inputs."""
import numpy as np
import pytest
import torch

from asr._lib import act_dtype as _act_dtype      # bfloat16, or float16 when the half build is under test (ASR_ACT=f16)

from oracle import nn as onn

pytestmark = pytest.mark.gpu

BF16, F32 = _act_dtype(), torch.float32


def _bf(t):
    """round an f32 CPU tensor to bf16 precision (keeps f32 dtype)"""
    return t.to(BF16).to(F32)


def _rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("M,N,K", [(200, 96, 72), (129, 130, 64), (1000, 1536, 384), (64, 3000, 320), (4096, 640, 512),
                                   (300, 64, 119), (77, 29, 45)])
@pytest.mark.parametrize("out_dtype", [BF16, F32])
def test_gemm_nt(device, M, N, K, out_dtype):
    from asr import _ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g))
    b = _bf(torch.randn(N, K, generator=g))
    bias = torch.randn(N, generator=g)
    ref = a.double() @ b.double().T + bias.double()
    out = _ops.gemm_nt(a.to(device, BF16), b.to(device, BF16), bias.to(device), out_dtype).cpu().float()
    tol = 1e-5 if out_dtype == F32 else 4e-3
    assert _rel(out, ref) < tol
    out2 = _ops.gemm_nt(a.to(device, BF16), b.to(device, BF16), None, F32).cpu()
    assert _rel(out2, a.double() @ b.double().T) < 1e-5


@pytest.mark.parametrize("M,N,K,bias_on", [(8000, 3072, 512, True), (7777, 3000, 320, True), (8192, 4100, 64, False), (5000, 6400, 96, True),
                                           (32000, 320, 3000, False), (8192, 4100, 72, True), (5000, 6400, 40, False)])
@pytest.mark.parametrize("out_dtype", [BF16, F32])
def test_gemm_nt_many_tiles(device, M, N, K, bias_on, out_dtype):
    """>= 1024 tiles of 256 x 128: the persistent kernel (several tiles per workgroup, K pipeline across tile boundaries, register
    epilogue); ragged M, N not a multiple of the tile or of 16, one K step only; K a multiple of 8 but not of the K step (the last
    step is fetched short: K = 3000 is the logit gradient's shape)"""
    from asr import _ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf(torch.randn(M, K, generator=g))
    b = _bf(torch.randn(N, K, generator=g))
    bias = torch.randn(N, generator=g) if bias_on else None
    out = _ops.gemm_nt(a.to(device, BF16), b.to(device, BF16), None if bias is None else bias.to(device), out_dtype)
    ref = torch.addmm(bias.to(device), a.to(device), b.to(device).T) if bias_on else a.to(device) @ b.to(device).T      # f32 on the GPU
    tol = 2e-5 if out_dtype == F32 else 4e-3
    assert _rel(out.float().cpu(), ref.cpu()) < tol
    assert (out.float() - ref).abs().max().item() < (1e-3 if out_dtype == F32 else 0.02 * ref.abs().max().item())      # no stray tile


@pytest.mark.parametrize("M,N,K,bias_on,lda", [(256, 256, 64, False, None), (300, 260, 192, True, None), (1000, 384, 320, False, None), (777, 132, 448, True, None),
                                                (513, 1024, 1024, True, 1088), (255, 4, 64, False, None), (8000, 512, 3072, True, None), (4096, 1280, 1024, False, None),
                                                (32000, 3072, 512, True, None), (700, 264, 72, True, None), (4000, 320, 3000, False, None), (300, 12, 200, True, None)])
@pytest.mark.parametrize("out_dtype", [BF16, F32])
def test_gemm_nt_8ph(device, M, N, K, bias_on, lda, out_dtype):
    """asr_gemm_nt_8ph (csrc/gemm8.hip: 256 x 256 tile, eight waves in two staggered groups, four half tiles of LDS-DMA in flight) called
    directly: one K step, an odd number of K steps, K not a multiple of 64 (the logit gradient's 3000), ragged M and N (a last column tile
    of 4 or 12), strided rows, bias, both output types --
    against the float32 product of the same bf16 operands on the GPU"""
    from asr import _ops
    g = torch.Generator().manual_seed(M + N + K)
    a = _bf(torch.randn(M, lda or K, generator=g)).to(device, BF16)[:, :K]
    b = _bf(torch.randn(N, K, generator=g)).to(device, BF16)
    bias = torch.randn(N, generator=g).to(device) if bias_on else None
    out = _ops.gemm_nt_8ph(a, b, bias, out_dtype)
    ref = a.float() @ b.float().T
    if bias_on:
        ref = ref + bias
    tol = 2e-5 if out_dtype == F32 else 4e-3
    assert _rel(out.float().cpu(), ref.cpu()) < tol
    assert (out.float() - ref).abs().max().item() < (1e-3 if out_dtype == F32 else 0.02 * ref.abs().max().item())      # no stray tile


def test_gemm_nt_8ph_exact_and_repeatable(device):
    """small integers are exact in bf16 and float32: any mix-up of rows, columns or K slices between the loader's permuted LDS rows and the
    epilogue's column map shows as a wrong integer; and twenty launches on large operands give the same bits every time (a fragment read
    that overtakes its LDS-DMA shows up as a tile that differs between runs: the guide's warning about this schedule)"""
    from asr import _ops
    g = torch.Generator().manual_seed(1)
    M, N, K = 700, 516, 320
    a = torch.randint(-4, 5, (M, K), generator=g).float()
    b = torch.randint(-4, 5, (N, K), generator=g).float()
    out = _ops.gemm_nt_8ph(a.to(device, BF16), b.to(device, BF16), None, F32).cpu()
    assert torch.equal(out, a @ b.T)
    eye = torch.eye(512)
    bb = (torch.arange(512 * 512, dtype=F32).reshape(512, 512) % 251 - 100)
    assert torch.equal(_ops.gemm_nt_8ph(eye.to(device, BF16), bb.to(device, BF16), None, F32).cpu(), bb.T.contiguous())
    a = torch.randn(16384, 2048, generator=g).to(device, BF16)
    b = torch.randn(2048, 2048, generator=g).to(device, BF16)
    first = _ops.gemm_nt_8ph(a, b, None, BF16).clone()
    scratch = torch.empty(1 << 26, device=device)
    for i in range(20):
        if i % 4 == 0:
            scratch.fill_(float(i))            # other traffic between the launches: different arrival orders of the half tiles
        assert torch.equal(_ops.gemm_nt_8ph(a, b, None, BF16), first)
    assert _rel(first.float().cpu(), (a.float() @ b.float().T).cpu()) < 4e-3


@pytest.mark.parametrize("M,N,K,bias_on,lda", [(32000, 3000, 320, True, None), (20037, 1100, 192, True, None), (16640, 4100, 128, False, None),
                                                (70000, 260, 256, True, 320), (32000, 3072, 384, True, None), (9000, 8192, 640, True, None)])
@pytest.mark.parametrize("out_dtype", [BF16, F32])
def test_gemm_nt_8ph_persistent_form(device, M, N, K, bias_on, lda, out_dtype):
    """more tiles than CUs and K <= 1024: asr_gemm_nt_8ph runs gemm_nt_8pp_kernel (one workgroup per CU walks its tiles as ONE stream of K
    steps, stores in the middle of it): odd and even numbers of K steps per tile (the LDS parity then flips from tile to tile), ragged last
    row and column tiles, workgroups with one tile more than others (395 tiles), strided rows, the bias out of LDS up to N = 8192;
    small integers make every output exact, so a K step taken from the neighbouring tile or a store of a half-cleared accumulator shows"""
    from asr import _ops
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randint(-3, 4, (M, lda or K), generator=g).float().to(device, BF16)[:, :K]
    b = torch.randint(-3, 4, (N, K), generator=g).float().to(device, BF16)
    bias = torch.randint(-8, 9, (N,), generator=g).float().to(device) if bias_on else None
    ref = a.float() @ b.float().T
    if bias_on:
        ref = ref + bias
    # five launches into an output pre-filled with NaN: every element written, and written with ITS value every time (a 16-byte buffer
    # store whose first data register the next instruction overwrote stored an address in lanes 12 - 15 of a row now and then -- 400 of
    # 24.6 M elements, one launch in a few: ASR8_STORE_FENCE in csrc/gemm8.hip)
    for _ in range(5):
        out = torch.full((M, N), float("nan"), dtype=out_dtype, device=device)
        _ops.gemm_nt_8ph(a, b, bias, out_dtype, out)
        assert torch.equal(out, ref if out_dtype == F32 else ref.to(BF16))
    # and on random operands, twice: the same bits, the float32 product within rounding
    a = torch.randn(M, K, generator=g).to(device, BF16)
    b = torch.randn(N, K, generator=g).to(device, BF16)
    first = _ops.gemm_nt_8ph(a, b, bias, out_dtype).clone()
    torch.empty(1 << 24, device=device).fill_(1.0)
    assert torch.equal(_ops.gemm_nt_8ph(a, b, bias, out_dtype), first)
    ref = a.float() @ b.float().T + (bias if bias_on else 0.0)
    assert _rel(first.float().cpu(), ref.cpu()) < (2e-5 if out_dtype == F32 else 4e-3)


def test_persistent_nt_kernel_with_float32_output_in_a_forced_process():
    """float32 products go to the one-tile kernel by default (bound by their stores either way); ASR_DEBUG nt8pp_f32=1 (read once per
    process) runs the persistent form on them: the float32 cases of the test above through gemm_nt_8pp_kernel<float>"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ASR_DEBUG="nt8pp_f32=1")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-m", "gpu", "-x", "-q",
                          "-k", "test_gemm_nt_8ph_persistent_form and out_dtype1"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_nt_kernels_without_the_8ph_kernel_in_a_forced_process():
    """asr_gemm_nt routes qualifying products to the 256 x 256 / eight-wave kernel; ASR_DEBUG nt_8ph=0 (read once per process) keeps the
    kernels it replaced (persistent 256 x 128, 128 x 128) under the NT tests of this file"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ASR_DEBUG="nt_8ph=0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-m", "gpu", "-x", "-q",
                          "-k", "test_gemm_nt and not 8ph"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_gemm_nt_asymmetric_identity(device):
    """A = I with an asymmetric B catches a transposed accumulator write (row/col swap)."""
    from asr import _ops
    n = 128
    a = torch.eye(n)
    b = torch.arange(n * n, dtype=F32).reshape(n, n) % 251 - 100     # exactly representable in bf16
    out = _ops.gemm_nt(a.to(device, BF16), b.to(device, BF16), None, F32).cpu()
    assert torch.equal(out, b.T.contiguous())


@pytest.mark.parametrize("K,M,N", [(1000, 130, 70), (333, 48, 960), (4096, 1536, 512), (2048, 3000, 320)])
def test_gemm_tn_acc(device, K, M, N):
    from asr import _ops
    g = torch.Generator().manual_seed(K + M + N)
    a = _bf(torch.randn(K, M, generator=g))
    b = _bf(torch.randn(K, N, generator=g))
    c0 = torch.randn(M, N, generator=g)
    ref = c0.double() + a.double().T @ b.double()
    c = c0.to(device)
    _ops.gemm_tn_acc(a.to(device, BF16), b.to(device, BF16), c)
    assert _rel(c.cpu(), ref) < 1e-5


@pytest.mark.parametrize("K,M,N,lda", [(64, 256, 256, None), (96, 40, 24, None), (333, 48, 960, None), (1000, 136, 72, None), (4096, 1536, 512, None),
                                        (2048, 3000, 320, None), (5000, 264, 520, 320), (20000, 640, 512, None)])
def test_gemm_tn_8ph(device, K, M, N, lda):
    """asr_gemm_tn_acc_group_8ph (csrc/gemm8.hip: eight waves, transposing LDS reads, split K with float atomics) called directly: one K
    step, K not a multiple of 64 (the rows beyond K arrive as zeros), ragged M and N, a strided operand -- small integers give EXACT sums
    (any mix-up of k slots between the A and B fragments, or of columns in the epilogue, is a wrong integer), random data the float32 bar"""
    from asr import _ops
    g = torch.Generator().manual_seed(K + M + N)
    a = torch.randint(-4, 5, (K, lda or M), generator=g).float()
    b = torch.randint(-4, 5, (K, N), generator=g).float()
    c = torch.full((M, N), 3.0, device=device)
    _ops.gemm_tn_acc_group_8ph([(a.to(device, BF16)[:, :M], b.to(device, BF16), c)])
    assert torch.equal(c.cpu(), 3.0 + a[:, :M].T @ b)
    a = _bf(torch.randn(K, lda or M, generator=g))
    b = _bf(torch.randn(K, N, generator=g))
    c0 = torch.randn(M, N, generator=g)
    c = c0.to(device)
    _ops.gemm_tn_acc_group_8ph([(a.to(device, BF16)[:, :M], b.to(device, BF16), c)])
    assert _rel(c.cpu(), c0.double() + a[:, :M].double().T @ b.double()) < 1e-5


def test_gemm_tn_8ph_group_repeatable(device):
    """the three weight gradients of a GRU layer in one launch, five times over: the float atomics of the K splits may land in any order
    (last-bit differences), a fragment read overtaking its LDS-DMA would be a wrong tile: every run within the float32 bar of the others and
    of the products formed one by one"""
    from asr import _ops
    T, B, H = 250, 32, 512
    g = torch.Generator().manual_seed(7)
    dgi = torch.randn(T * B, 6 * H, generator=g).to(device, BF16)
    x = torch.randn(T * B, H, generator=g).to(device, BF16)
    dgh = torch.randn(T * B, 6 * H, generator=g).to(device, BF16)
    h16 = torch.randn(T * B, 2 * H, generator=g).to(device, BF16)
    ref_ih = (dgi.float().T @ x.float())
    ref_hh = [dgh[B:, :3 * H].float().T @ h16[:-B, :H].float(), dgh[:-B, 3 * H:].float().T @ h16[B:, H:].float()]
    scratch = torch.empty(1 << 26, device=device)
    for it in range(5):
        dwih = torch.zeros(6 * H, H, device=device)
        dwhh = torch.zeros(2, 3 * H, H, device=device)
        scratch.fill_(float(it))
        _ops.gemm_tn_acc_group_8ph([(dgi, x, dwih), (dgh[B:, :3 * H], h16[:-B, :H], dwhh[0]), (dgh[:-B, 3 * H:], h16[B:, H:], dwhh[1])])
        assert _rel(dwih.cpu(), ref_ih.cpu()) < 1e-5
        assert _rel(dwhh[0].cpu(), ref_hh[0].cpu()) < 1e-5 and _rel(dwhh[1].cpu(), ref_hh[1].cpu()) < 1e-5


def test_tn_kernels_without_the_8ph_kernel_in_a_forced_process():
    """asr_gemm_tn_acc / asr_gemm_tn_acc_group route qualifying products to the eight-wave kernel; ASR_DEBUG tn_8ph=0 (read once per process)
    keeps the kernels it replaced under the TN tests of this file"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ASR_DEBUG="tn_8ph=0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-m", "gpu", "-x", "-q",
                          "-k", "test_gemm_tn and not 8ph"], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_gemm_tn_exact_integers(device):
    """small integers: exact in bf16 and f32, so any k-slot mix-up between A and B shows as a wrong integer."""
    from asr import _ops
    g = torch.Generator().manual_seed(0)
    K, M, N = 96, 40, 24
    a = torch.randint(-4, 5, (K, M), generator=g).float()
    b = torch.randint(-4, 5, (K, N), generator=g).float()
    c = torch.zeros(M, N, device=device)
    _ops.gemm_tn_acc(a.to(device, BF16), b.to(device, BF16), c)
    assert torch.equal(c.cpu(), a.T @ b)


def test_gemm_tn_strided_views(device):
    from asr import _ops
    g = torch.Generator().manual_seed(3)
    K, M, N = 500, 96, 64
    big_a = _bf(torch.randn(K, 3 * M, generator=g))
    big_b = _bf(torch.randn(K, 2 * N, generator=g))
    c = torch.zeros(M, N, device=device)
    _ops.gemm_tn_acc(big_a.to(device, BF16)[:, M:2 * M], big_b.to(device, BF16)[:, N:], c)
    ref = big_a[:, M:2 * M].double().T @ big_b[:, N:].double()
    assert _rel(c.cpu(), ref) < 1e-5


def test_gemm_tn_group_at_model_size(device):
    """the three weight-gradient products a BASELINE GRU layer releases together (3072 x 512 and twice 1536 x 512 over K = 32000
    rows, on the shifted strided views of _GRU.backward) in one grouped launch, against float32 products of the same operands on
    the device, with and without the vector kernel's fast path being possible (an odd leading dimension forces the general kernel)"""
    from asr import _ops
    g = torch.Generator(device=device).manual_seed(9)
    T, B, H, I = 1000, 32, 512, 512
    K = T * B
    for pad in (0, 1):                                           # pad = 1: lda / ldb not multiples of 8 -> general kernel
        dgi = torch.randn(K, 6 * H + pad, generator=g, device=device).to(BF16)[:, :6 * H]
        dgh = torch.randn(K, 6 * H + pad, generator=g, device=device).to(BF16)[:, :6 * H]
        x = torch.randn(K, I + pad, generator=g, device=device).to(BF16)[:, :I]
        h = torch.randn(K, 2 * H + pad, generator=g, device=device).to(BF16)[:, :2 * H]
        c = [torch.zeros(6 * H, I, device=device), torch.zeros(3 * H, H, device=device), torch.zeros(3 * H, H, device=device)]
        prods = [(dgi, x, c[0]), (dgh[B:, :3 * H], h[:-B, :H], c[1]), (dgh[:-B, 3 * H:], h[B:, H:], c[2])]
        _ops.gemm_tn_acc_group(prods)
        for (a, b, got) in prods:
            ref = a.float().T @ b.float()
            assert _rel(got.cpu(), ref.cpu()) < 2e-5, pad


def test_gemm_tn_operand_beyond_2_gb(device):
    """the vector TN kernel addresses its operands with 32-bit byte offsets (buffer loads, the k offset in an SGPR): an operand whose
    rows lie 64 KB apart reaches byte 2.6 G at K = 40000 -- every offset beyond 2^31 must still be the right one"""
    from asr import _ops
    g = torch.Generator(device=device).manual_seed(5)
    K, M, N, ld = 40000, 128, 64, 32768
    big = torch.randn(K, ld, generator=g, device=device, dtype=F32).to(BF16)
    a = big[:, 256:256 + M]
    b = torch.randn(K, N, generator=g, device=device, dtype=F32).to(BF16)
    c = torch.zeros(M, N, device=device)
    _ops.gemm_tn_acc(a, b, c)
    ref = a.double().T @ b.double()
    assert _rel(c.cpu(), ref.cpu()) < 1e-5
    last = torch.zeros(M, N, device=device)                  # the last k rows alone (offsets ~2.6 G): a wrapped offset reads other rows
    _ops.gemm_tn_acc(a[K - 100:], b[K - 100:], last)
    assert _rel(last.cpu(), (a[K - 100:].double().T @ b[K - 100:].double()).cpu()) < 1e-5


@pytest.mark.parametrize("K,M,N,I,B", [(96, 40, 24, 16, 8), (2016, 1536, 512, 384, 32), (700, 130, 70, 264, 4)])
def test_gemm_tn_acc_group(device, K, M, N, I, B):
    """the grouped launch against float64 on the three products the GRU backward hands it: (K, 2M)^T (K, I), and per direction
    (K - B, M)^T (K - B, N) on shifted strided views (direction 0 pairs rows t with t - 1, direction 1 t with t + 1); small
    integers at the first shape (exact); two products into ONE output; grouped == one by one."""
    from asr import _ops
    g = torch.Generator().manual_seed(K + I)
    exact = K == 96
    draw = (lambda *s: torch.randint(-4, 5, s, generator=g).float()) if exact else (lambda *s: _bf(torch.randn(*s, generator=g)))
    dgi, x, dgh, h = draw(K, 2 * M), draw(K, I), draw(K, 2 * M), draw(K, 2 * N)
    c0 = [torch.zeros(s) if exact else torch.randn(*s, generator=g) for s in ((2 * M, I), (M, N), (M, N))]
    dev = lambda t: t.to(device, BF16)
    d_dgi, d_x, d_dgh, d_h = dev(dgi), dev(x), dev(dgh), dev(h)
    def views(a, hh):
        return [(a[B:, :M], hh[:-B, :N]), (a[:-B, M:], hh[B:, N:])]
    c = [t.to(device) for t in c0]
    prods = [(d_dgi, d_x, c[0])] + [(a, b, c[1 + d]) for d, (a, b) in enumerate(views(d_dgh, d_h))]
    _ops.gemm_tn_acc_group(prods)
    ref = [c0[0].double() + dgi.double().T @ x.double()] + [c0[1 + d].double() + a.double().T @ b.double()
                                                            for d, (a, b) in enumerate(views(dgh, h))]
    for got, want in zip(c, ref):
        if exact:
            assert torch.equal(got.cpu().double(), want)
        else:
            assert _rel(got.cpu(), want) < 1e-5
    one = [t.to(device) for t in c0]
    for (a, b, _), o in zip(prods, one):
        _ops.gemm_tn_acc(a, b, o)
    for got, o in zip(c, one):
        assert _rel(got.cpu(), o.cpu().double()) < 1e-6
    both = torch.zeros(M, N, device=device)                         # two products adding into one output
    (a0, b0), (a1, b1) = views(d_dgh, d_h)
    _ops.gemm_tn_acc_group([(a0, b0, both), (a1, b1, both)])
    assert _rel(both.cpu(), (ref[1] - c0[1].double()) + (ref[2] - c0[2].double())) < 1e-5


@pytest.mark.parametrize("B,Cin,Hin,T,Cout,pad_h,first", [(2, 3, 12, 37, 16, 0, True), (3, 8, 13, 50, 24, 0, False),
                                                          (2, 16, 9, 33, 32, 1, False)])
def test_conv_as_im2col_gemm(device, B, Cin, Hin, T, Cout, pad_h, first):
    """forward, input gradient (col2im) and weight gradient of the causal conv vs torch conv2d."""
    from asr import _ops
    g = torch.Generator().manual_seed(T)
    KH, KW = 3, 5
    x = _bf(torch.randn(B, Cin, Hin, T, generator=g))
    W = _bf(torch.randn(Cout, Cin, KH, KW, generator=g) * 0.2)
    bias = torch.randn(Cout, generator=g)
    xr = x.clone().requires_grad_(True)
    Wr = W.clone().requires_grad_(True)
    y_ref = onn.conv2d_causal(xr, Wr, bias, pad_h)
    Hout = y_ref.shape[2]
    gy = _bf(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(gy)
    if first:   # reference layout (B, C, H, T) f32 straight from the loader
        xd = x.to(device)
        strides = (xd.stride(3), xd.stride(0), xd.stride(2), xd.stride(1))
    else:       # internal layout (T, B, H, C) bf16
        xd = x.permute(3, 0, 2, 1).contiguous().to(device, BF16)
        strides = tuple(xd.stride())
    col = _ops.im2col(xd, strides, T, B, Hin, Cin, KH, KW, pad_h)
    Kp = col.shape[1]
    Wm = torch.zeros(Cout, Kp)
    Wm[:, :KH * KW * Cin] = W.permute(0, 2, 3, 1).reshape(Cout, -1)       # k = (kh, kw, ci)
    y = _ops.gemm_nt(col, Wm.to(device, BF16), bias.to(device), F32)      # rows (t, b, ho)
    y = y.reshape(T, B, Hout, Cout).permute(1, 3, 2, 0).cpu()
    assert _rel(y, y_ref.detach()) < 1e-5
    gyd = gy.permute(3, 0, 2, 1).reshape(T * B * Hout, Cout).contiguous().to(device, BF16)
    dW = torch.zeros(Cout, Kp, device=device)
    _ops.gemm_tn_acc(gyd, col, dW)
    dW = dW[:, :KH * KW * Cin].reshape(Cout, KH, KW, Cin).permute(0, 3, 1, 2).cpu()
    assert _rel(dW, Wr.grad) < 1e-5
    dcol = _ops.gemm_nt(gyd, Wm.T.contiguous().to(device, BF16), None, BF16)
    dx = _ops.col2im(dcol, T, B, Hin, Cin, KH, KW, pad_h).float().permute(1, 3, 2, 0).cpu()
    assert _rel(dx, xr.grad) < 6e-3       # dcol is rounded to bf16 before the 15-tap gather


def test_maxout_and_maxpool(device):
    from asr import _ops
    g = torch.Generator().manual_seed(9)
    B, C, H, T = 3, 16, 11, 21
    x = _bf(torch.randn(B, C, H, T, generator=g))
    xr = x.clone().requires_grad_(True)
    y_ref = onn.maxout2(xr)
    gy = _bf(torch.randn(y_ref.shape, generator=g))
    y_ref.backward(gy)
    xd = x.permute(3, 0, 2, 1).contiguous().to(device, BF16)
    y = _ops.maxout2_fwd(xd)
    assert torch.equal(y.float().permute(1, 3, 2, 0).cpu(), y_ref.detach())
    dx = _ops.maxout2_bwd(xd, gy.permute(3, 0, 2, 1).contiguous().to(device, BF16))
    assert torch.equal(dx.float().permute(1, 3, 2, 0).cpu(), xr.grad)
    for k, Hin in ((3, 38), (2, 11), (2, 4), (3, 11)):
        x = _bf(torch.randn(B, C, Hin, T, generator=g))
        xr = x.clone().requires_grad_(True)
        y_ref = onn.maxpool_h(xr, k)
        gy = _bf(torch.randn(y_ref.shape, generator=g))
        y_ref.backward(gy)
        xd = x.permute(3, 0, 2, 1).contiguous().to(device, BF16)
        y = _ops.maxpool_h_fwd(xd, k)
        assert y.shape[2] == y_ref.shape[2]
        assert torch.equal(y.float().permute(1, 3, 2, 0).cpu(), y_ref.detach())
        dx = _ops.maxpool_h_bwd(xd, gy.permute(3, 0, 2, 1).contiguous().to(device, BF16), k)
        assert torch.equal(dx.float().permute(1, 3, 2, 0).cpu(), xr.grad)


@pytest.mark.parametrize("B,C,H,T", [(5, 3, 40, 10), (2, 3000, 1, 7), (3, 64, 6, 9)])
def test_layernorm(device, B, C, H, T):
    """same shapes/ranges as the reference's own test (asr/nn/test_layernorm.py:30-43) plus the logit shape."""
    from asr import _ops
    rs = np.random.RandomState(C)
    x = rs.uniform(-10, 10, (B, C, H, T)).astype(np.float32)
    gy = rs.uniform(-1, 1, (B, C, H, T)).astype(np.float32)
    gamma = rs.uniform(0.5, 1.5, C).astype(np.float32)
    beta = rs.uniform(-1, 1, C).astype(np.float32)
    y_ref, cache = onn.layer_normalization(x.astype(np.float64), gamma.astype(np.float64), beta.astype(np.float64))
    dx_ref, dg_ref, db_ref = onn.layer_normalization_bwd(gy.astype(np.float64), gamma.astype(np.float64), cache)
    xd = torch.tensor(x).permute(3, 0, 2, 1).reshape(T * B, H * C).contiguous().to(device)
    gyd = torch.tensor(gy).permute(3, 0, 2, 1).reshape(T * B, H * C).contiguous().to(device)
    gd, bd = torch.tensor(gamma).to(device), torch.tensor(beta).to(device)
    y, mean, rstd = _ops.layernorm_fwd(xd, gd, bd, C, F32)
    y = y.reshape(T, B, H, C).permute(1, 3, 2, 0).cpu().numpy()
    np.testing.assert_allclose(y, y_ref, rtol=1e-3, atol=1e-3)       # tolerance of asr/nn/test_layernorm.py:45-50
    np.testing.assert_allclose(y, y_ref, rtol=1e-5, atol=2e-5)
    dgamma = torch.zeros(C, device=device)
    dbeta = torch.zeros(C, device=device)
    dx = _ops.layernorm_bwd(xd, gyd, gd, mean, rstd, C, F32, dgamma, dbeta)
    dx = dx.reshape(T, B, H, C).permute(1, 3, 2, 0).cpu().numpy()
    np.testing.assert_allclose(dx, dx_ref, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(dgamma.cpu().numpy(), dg_ref, rtol=1e-4, atol=1e-3)
    np.testing.assert_allclose(dbeta.cpu().numpy(), db_ref, rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("rows,D", [(37, 3000), (1000, 120), (5, 4096), (130, 1024), (64, 260)])
def test_layernorm_rows_in_registers_with_lse(device, rows, D):
    """float32 rows normalised over their whole width (the logits path): the wave-per-row forward against float64, plain and with the
    log-sum-exp of every output row (what asr_ctc_forward_lse takes over)"""
    from asr import _ops
    g = torch.Generator().manual_seed(rows + D)
    x = torch.randn(rows, D, generator=g) * 3 + 0.5
    gamma, beta = torch.rand(D, generator=g) + 0.5, torch.randn(D, generator=g)
    xd = x.double()
    mu = xd.mean(1, keepdim=True)
    var = ((xd - mu) ** 2).mean(1, keepdim=True)
    ref = (xd - mu) / var.sqrt() * gamma.double() + beta.double()
    y, mean, rstd = _ops.layernorm_fwd(x.to(device), gamma.to(device), beta.to(device), D, F32)
    y2, mean2, rstd2, lse = _ops.layernorm_fwd(x.to(device), gamma.to(device), beta.to(device), D, F32, want_lse=True)
    assert torch.equal(y, y2) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    assert _rel(y.cpu(), ref) < 2e-6
    assert torch.allclose(mean.cpu().double(), mu[:, 0], atol=1e-5) and torch.allclose(rstd.cpu().double(), 1 / var.sqrt()[:, 0], rtol=1e-5)
    assert torch.allclose(lse.cpu().double(), torch.logsumexp(ref, 1), rtol=0, atol=2e-5)


@pytest.mark.parametrize("rows,D,C", [(20000, 120, 120), (700, 3000, 3000), (33, 4096, 1024), (257, 1028, 4), (1, 8, 8)])
def test_layernorm_one_sweep_backward(device, rows, D, C):
    """asr_layernorm_bwd_rows (dx + column sums in one pass, many rows per workgroup) against the float64 oracle and,
    for the bf16 output, against the rounded float32 output; also dx only / parameter gradients only"""
    from asr import _ops, _lib
    assert _lib.lib().asr_layernorm_bwd_rows_ws_bytes(rows, D) > 0
    rs = np.random.RandomState(rows + D)
    H = D // C
    x = rs.uniform(-10, 10, (rows, H, C)).astype(np.float32)
    gy = rs.uniform(-1, 1, (rows, H, C)).astype(np.float32)
    gamma = rs.uniform(0.5, 1.5, C).astype(np.float32)
    beta = rs.uniform(-1, 1, C).astype(np.float32)
    # oracle layout: (B, C, H, T) with B = rows, T = 1
    x_o, gy_o = x.transpose(0, 2, 1)[..., None].astype(np.float64), gy.transpose(0, 2, 1)[..., None].astype(np.float64)
    _, cache = onn.layer_normalization(x_o, gamma.astype(np.float64), beta.astype(np.float64))
    dx_ref, dg_ref, db_ref = onn.layer_normalization_bwd(gy_o, gamma.astype(np.float64), cache)
    dx_ref = dx_ref[..., 0].transpose(0, 2, 1).reshape(rows, D)
    xd, gyd = torch.tensor(x).reshape(rows, D).to(device), torch.tensor(gy).reshape(rows, D).to(device)
    gd, bd = torch.tensor(gamma).to(device), torch.tensor(beta).to(device)
    _, mean, rstd = _ops.layernorm_fwd(xd, gd, bd, C, F32)
    dgamma, dbeta = torch.zeros(C, device=device), torch.zeros(C, device=device)
    dx = _ops.layernorm_bwd(xd, gyd, gd, mean, rstd, C, F32, dgamma, dbeta)
    np.testing.assert_allclose(dx.cpu().numpy(), dx_ref, rtol=1e-4, atol=1e-5)
    scale = max(1.0, float(np.sqrt(rows * H)))
    np.testing.assert_allclose(dgamma.cpu().numpy(), dg_ref, rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(dbeta.cpu().numpy(), db_ref, rtol=1e-4, atol=1e-5 * scale)
    dx16 = _ops.layernorm_bwd(xd, gyd, gd, mean, rstd, C, BF16, None, None)        # dx only, bf16
    # (the two instantiations contract their multiply-adds differently: equal up to one bf16 rounding step)
    assert dx16.dtype == BF16
    torch.testing.assert_close(dx16.float(), dx, rtol=2.0 ** -7, atol=1e-6)
    dg2, db2 = torch.ones(C, device=device), torch.ones(C, device=device)          # parameter gradients only, accumulated
    assert _ops.layernorm_bwd(xd, gyd, gd, mean, rstd, C, F32, dg2, db2, need_dx=False) is None
    np.testing.assert_allclose(dg2.cpu().numpy() - 1.0, dgamma.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale)
    np.testing.assert_allclose(db2.cpu().numpy() - 1.0, dbeta.cpu().numpy(), rtol=1e-4, atol=1e-4 * scale)


@pytest.mark.parametrize("T,B,H,C,k", [(7, 3, 38, 64, 3), (5, 2, 13, 16, 2), (4, 2, 11, 24, 3), (3, 1, 4, 8, 2), (6, 2, 6, 32, 1),
                                       (300, 16, 38, 64, 3), (9, 4, 13, 128, 2), (9, 4, 13, 256, 2), (40, 8, 6, 256, 3), (5, 3, 4, 512, 2)])
def test_fused_maxout_pool_equals_the_two_kernels(device, T, B, H, C, k):
    """asr_maxout2_pool_fwd / _bwd against asr_maxout2_* followed by asr_maxpool_h_* (cover_all windows, ties included):
    bit-identical outputs and input gradients.  C = 256 / 512: the bias-gradient fold needs more than one round of the workgroup's
    threads (half of db was dropped at C = 256)"""
    from asr import _ops
    g = torch.Generator().manual_seed(T * 100 + H)
    x = torch.randn(T, B, H, 2 * C, generator=g)
    x = (x * 2).round() / 2                     # many exact ties inside pairs and windows
    xd = x.to(device).to(BF16)
    assert _ops.maxout2_pool_ok(xd)
    y_ref = _ops.maxpool_h_fwd(_ops.maxout2_fwd(xd), k)
    y = _ops.maxout2_pool_fwd(xd, k)
    assert y.shape == y_ref.shape and torch.equal(y, y_ref)
    gy = torch.randn(y.shape, generator=g).to(device).to(BF16)
    mid = _ops.maxout2_fwd(xd)
    dx_ref = _ops.maxout2_bwd(xd, _ops.maxpool_h_bwd(mid, gy, k))
    dx = _ops.maxout2_pool_bwd(xd, gy, k)
    assert torch.equal(dx, dx_ref)
    # with the bias gradient of the producer: the column sums of dx, accumulated on top of what is there
    if _ops.maxout2_pool_bwd_db_ok(C):
        db = torch.ones(2 * C, device=device)
        dx2 = _ops.maxout2_pool_bwd(xd, gy, k, db)
        assert torch.equal(dx2, dx_ref)
        cols = dx_ref.double().reshape(-1, 2 * C)
        ref = 1.0 + cols.sum(0)
        # float32 atomics in launch order: exact for bf16 gradients of this size, rounding for f16 ones (ASR_ACT=f16: 11-bit
        # mantissas, 182400 rows) -- bounded by a few ulp of the column's sum of magnitudes; a dropped round is off by the sum itself
        tol = 1e-4 + 2e-7 * float(cols.abs().sum(0).max())
        assert float((db.double() - ref).abs().max()) <= tol
    else:
        assert C == 24


def test_layer_stack_fuses_maxout_and_pooling(device):
    """nn containers run Maxout(2) [+ Dropout(0)] + MaxPooling2D((k, 1)) as one pass: same output and gradient as the layers
    applied one by one"""
    from asr import nn, functions as F
    from asr.nn.nn import _fusable_pool
    torch.manual_seed(2)
    x = torch.randn(2, 32, 13, 9).to(device).to(BF16)      # logical (B, 2C, H, T)
    layers = [nn.Maxout(2), nn.Dropout(0), nn.MaxPooling2D(ksize=(3, 1))]
    assert _fusable_pool(layers, 0) == 2
    xa = x.clone().requires_grad_(True)
    ya = nn.Module(*layers)(xa)
    xb = x.clone().requires_grad_(True)
    yb = F.max_pooling_2d(F.maxout(xb, 2), (3, 1))
    assert ya.shape == yb.shape and torch.equal(ya, yb)
    gy = torch.randn_like(ya)
    ya.backward(gy)
    yb.backward(gy)
    assert torch.equal(xa.grad, xb.grad)
    assert _fusable_pool([nn.Maxout(2), nn.Dropout(0.5), nn.MaxPooling2D(ksize=(3, 1))], 0) == -1
    assert _fusable_pool([nn.Maxout(2), nn.MaxPooling2D(ksize=(3, 1), stride=2)], 0) == -1


def test_conv_bias_gradient_from_the_fused_maxout_pool_backward(device):
    """Convolution2D -> Maxout(2) -> MaxPooling2D in one layer stack: the fused backward forms the convolution's bias gradient (column
    sums of the gradient it scatters); it must equal what the convolution's own backward computes when the layers run one by one,
    also on a later pass over a fresh graph"""
    from asr import nn, functions as F
    from asr import _ops
    torch.manual_seed(5)
    x = torch.randn(3, 3, 20, 31, device=device)                      # (B, C, H, T) float32 input of a first layer
    conv = nn.Convolution2D(3, 32, (3, 5), stride=1, pad=(0, 4), causal=True).to(device)
    stack = nn.Module(conv, nn.Maxout(2), nn.Dropout(0), nn.MaxPooling2D(ksize=(3, 1)))
    grads = []
    F.BIAS_FROM_POOL_MIN_NUMEL[0], keep = 0, F.BIAS_FROM_POOL_MIN_NUMEL[0]       # (the layer stack fuses on large outputs only)
    for fused in (True, False, True):
        conv.cleargrads()
        y = stack(x) if fused else F.max_pooling_2d(F.maxout(conv(x), 2), (3, 1))
        gy = torch.ones_like(y) * 0.5
        y.backward(gy)
        F.join_side_stream()                # the weight-gradient products run on the side stream (the optimisers join it)
        grads.append((conv.b.grad.clone(), conv.W.grad.clone()))
    F.BIAS_FROM_POOL_MIN_NUMEL[0] = keep
    for gb, gW in grads[1:]:
        assert torch.allclose(gb, grads[0][0], rtol=1e-4, atol=1e-3) and torch.allclose(gW, grads[0][1], rtol=1e-4, atol=1e-3)
    assert grads[0][0].abs().max().item() > 0


# 1: one launch per time step, 2: persistent, placement-free hand-off,
# 4: persistent with the XCD-local hand-off where placement allows, signalled by flags; 8: the same with the payload as its
# own signal (= automatic); 7: 4 with a forged split placement
# 9: backward with the partial-sum exchange (bwd_ps_kernel; H % 128 == 0, else the wide kernel serves); 10: 9 with a forged split placement
@pytest.mark.parametrize("mode", [1, 2, 4, 7, 8, 9, 10])
@pytest.mark.parametrize("T,B,I,H,ndir", [(12, 4, 64, 64, 2), (9, 5, 96, 128, 1), (20, 32, 64, 512, 2), (7, 40, 32, 64, 2),
                                          (150, 32, 32, 256, 2), (40, 19, 48, 128, 2), (30, 7, 32, 384, 1)])
def test_gru_step_kernels(device, T, B, I, H, ndir, mode):
    """forward states and all gradients of the (Bi)GRU against torch.nn.GRU on CPU (weights rounded to bf16)."""
    from asr import _ops, _lib
    if mode >= 2 and B > 32:
        pytest.skip("an explicitly selected persistent form covers B <= 32 (mode 0 runs larger batches as slabs: the test below)")
    _ops.GRU_MODE[0] = mode
    try:
        _gru_case(device, T, B, I, H, ndir)
        _ops.gru_check_sync()
        if _lib.lib().asr_gru_fwd_accepts_bf16_gi(T, B, H, ndir, mode):      # the same with the input projections in bf16
            _gru_case(device, T, B, I, H, ndir, gi_dtype=BF16,
                      tol=dict(y=1e-2, dx=3e-2, dwih=3e-2, dbih=3e-2, dbhh=3e-2, dwhh=3e-2))
            _ops.gru_check_sync()
    finally:
        _ops.GRU_MODE[0] = 0


def test_gru_full_size_against_fp32_oracle(device):
    """the DEFAULT launch forms (persistent forward, wide backward) at BASELINE size T=1000, B=32, H=512,
    both directions, against torch.nn.GRU on the CPU in float32 (oracle.nn.bigru_sum) -- y, dx, dW_ih, dW_hh and both bias
    gradients.  1000 dependent steps exchange h in bf16; the stated tolerances are relative L2 errors."""
    from asr import _ops
    assert _ops.GRU_MODE[0] == 0
    errs = _gru_case(device, 1000, 32, 384, 512, 2, tol=dict(y=5e-3, dx=6e-3, dwih=6e-3, dbih=6e-3, dbhh=6e-3, dwhh=6e-3))   # measured: 1.0e-3 .. 2.0e-3
    _ops.gru_check_sync()
    print("full-size GRU vs fp32 oracle, relative L2 errors:", {k: "%.2e" % v for k, v in errs.items()})
    # what the model runs: the input projections written in bf16 by the projection GEMM
    assert _ops.gru_gi_dtype(1000, 32, 512, 2) == BF16
    errs = _gru_case(device, 1000, 32, 384, 512, 2, gi_dtype=BF16, tol=dict(y=8e-3, dx=1e-2, dwih=1e-2, dbih=1e-2, dbhh=1e-2, dwhh=1e-2))
    _ops.gru_check_sync()
    print("full-size GRU, bf16 input projections, vs fp32 oracle:", {k: "%.2e" % v for k, v in errs.items()})
    # the saved gates: IEEE half, blocked by workgroup (the default where asr_gru_gates_f16_ok) against float32 (switch off) -- same bars
    assert _ops.gru_gates_f16(1000, 32, 512, 2)
    _ops.GRU_GATES_F16[0] = False
    try:
        errs32 = _gru_case(device, 1000, 32, 384, 512, 2, gi_dtype=BF16, tol=dict(y=8e-3, dx=1e-2, dwih=1e-2, dbih=1e-2, dbhh=1e-2, dwhh=1e-2))
        _ops.gru_check_sync()
    finally:
        _ops.GRU_GATES_F16[0] = True
    print("full-size GRU, float32 saved gates (switch off), vs fp32 oracle:", {k: "%.2e" % v for k, v in errs32.items()})
    for k in errs:
        assert errs[k] < 1.25 * errs32[k] + 2e-4, (k, errs[k], errs32[k])       # half gates cost (next to) nothing in accuracy


@pytest.mark.parametrize("T,B,I,H,ndir", [(40, 48, 64, 128, 2), (24, 64, 48, 256, 2), (12, 128, 32, 128, 2), (30, 33, 32, 384, 1), (25, 77, 32, 128, 2)])
def test_gru_batches_beyond_the_resident_limit_run_as_slabs(device, T, B, I, H, ndir):
    """the reference trains with 128 utterances per bucket, shrinking by 16 (run/ctc/cnn/train.py:72-73,165,208-209);
    the persistent kernels are resident at 32.  asr_gru_fwd / asr_gru_bwd run a larger batch as consecutive slabs of <= 32 rows through
    the default kernel pair (csrc/gru.hip: slab_rows).  Utterances are independent, so (a) the results are those of torch.nn.GRU on the
    whole batch, (b) rows [32 k, 32 k + 32) of every output equal, BIT FOR BIT, a separate run on those utterances alone -- and the
    call takes the fast path (bf16 input projections and half gates are accepted, which only the default pair does)."""
    from asr import _ops, _lib
    assert _ops.GRU_MODE[0] == 0
    fast = H % 128 == 0                 # (the partial-sum backward kernel's condition; other widths keep the per-step launches beyond 32 rows)
    assert _lib.lib().asr_gru_fwd_accepts_bf16_gi(T, B, H, ndir, 0) == (1 if fast else 0)
    assert _lib.lib().asr_gru_gates_f16_ok(T, B, H, ndir, 0) == (1 if fast and H >= 256 else 0)     # (half gates: the forward ring form, K split >= 2)
    _gru_case(device, T, B, I, H, ndir)
    _ops.gru_check_sync()
    if H % 128:
        return              # (the backward partial-sum kernel needs H % 128 == 0: such a shape keeps the per-step launches)
    g = torch.Generator().manual_seed(B * H)
    k = 1.0 / np.sqrt(H)
    gi = (torch.randn(T, B, ndir * 3 * H, generator=g) * 0.7).to(device, BF16)
    whh = _bf(torch.empty(ndir, 3 * H, H).uniform_(-k, k, generator=g)).to(device, BF16).contiguous()
    whhT = whh.transpose(1, 2).contiguous()
    bhh = torch.empty(ndir * 3 * H).uniform_(-k, k, generator=g).to(device)
    gy = torch.randn(T, B, H, generator=g).to(device, BF16)
    x_len = torch.randint(T // 2, T + 1, (B,), generator=g, dtype=torch.int32)
    x_len[0] = T
    for lens in (None, x_len.to(device)):
        def run(rows):
            n = rows.stop - rows.start
            gis = gi[:, rows].contiguous().reshape(T * n, -1).clone()          # (the call pins the update gate of dead rows in place)
            if not _lib.lib().asr_gru_fwd_accepts_bf16_gi(T, n, H, ndir, 0):
                gis = gis.float()                                              # (same values: the wide kernel of a <= 16-row batch reads float32)
            ln = None if lens is None else lens[rows].contiguous()
            y, hseq, hseq16, gates = _ops.gru_fwd(gis, whh, bhh, T, n, H, ndir, ln)
            dbi, dbh = torch.zeros(ndir * 3 * H, device=device), torch.zeros(ndir * 3 * H, device=device)
            dgi, dgh = _ops.gru_bwd(gy[:, rows].contiguous().reshape(T * n, H), gates, hseq, whhT, T, n, H, ndir, dbi, dbh, ln)
            _ops.gru_check_sync()
            gs = _ops.gru_gates_standard(gates, H)
            return [t.reshape(T, n, -1) for t in (y, hseq, hseq16, gs, dgi, dgh)], dbi, dbh
        whole, dbi, dbh = run(slice(0, B))
        sum_i, sum_h = torch.zeros_like(dbi), torch.zeros_like(dbh)
        for b0 in range(0, B, 32):
            rows = slice(b0, min(B, b0 + 32))
            part, pi, ph = run(rows)
            n = rows.stop - rows.start
            # a slab of the batch goes through the 16-unit x 8-row / partial-sum pair; alone, a batch of <= 16 utterances takes the wide
            # forward kernel (another summation order): bit-exact where the same kernels serve, to rounding otherwise
            same_kernels = n > 16
            for name, a, b in zip(("y", "hseq", "hseq16", "gates", "dgi", "dgh"), whole, part):
                if name == "hseq16":        # (operand of the dW_hh product: h_{t-1} pairs with step t, the LAST step of a direction is never read -- nor written by the ring form)
                    a = torch.cat([a[:-1, :, :H], a[1:, :, H:]], dim=2) if ndir == 2 else a[:-1]
                    b = torch.cat([b[:-1, :, :H], b[1:, :, H:]], dim=2) if ndir == 2 else b[:-1]
                if same_kernels:
                    assert torch.equal(a[:, rows], b), (name, b0, lens is not None)
                else:
                    assert _rel(a[:, rows].float().cpu(), b.float().cpu()) < 2e-2, (name, b0, lens is not None)
            sum_i += pi
            sum_h += ph
        assert _rel(dbi.cpu(), sum_i.cpu()) < 5e-4 and _rel(dbh.cpu(), sum_h.cpu()) < 5e-4      # (float32 summation order; a <= 16-row remainder alone sums through asr_colsum_acc)


_GRU_REF = {}


def _gru_case(device, T, B, I, H, ndir, tol=None, gi_dtype=F32):
    from asr import _ops
    tol = tol or dict(y=6e-3, dx=2e-2, dwih=2e-2, dbih=2e-2, dbhh=2e-2, dwhh=2e-2)
    errs = {}

    def gate(name, a, b):
        errs[name] = max(errs.get(name, 0.0), _rel(a, b))
        assert errs[name] < tol[name], (name, errs[name], tol[name])
    key = (T, B, I, H, ndir)
    if key not in _GRU_REF:           # (the float32 CPU reference of a shape is the same for every device variant tested on it)
        g = torch.Generator().manual_seed(T * H)
        k = 1.0 / np.sqrt(H)
        P = dict(w_ih=_bf(torch.empty(ndir, 3 * H, I).uniform_(-k, k, generator=g)),
                 w_hh=_bf(torch.empty(ndir, 3 * H, H).uniform_(-k, k, generator=g)),
                 b_ih=torch.empty(ndir, 3 * H).uniform_(-k, k, generator=g),
                 b_hh=torch.empty(ndir, 3 * H).uniform_(-k, k, generator=g))
        x = _bf(torch.randn(T, B, I, generator=g))
        xr = x.clone().requires_grad_(True)
        y_ref, ref = onn.bigru_sum(xr, P, H, ndir)
        gy = _bf(torch.randn(T, B, H, generator=g))
        y_ref.backward(gy)
        if T * B * H >= 1 << 22:      # keep only what is expensive to make
            _GRU_REF.clear()
            _GRU_REF[key] = (P, x, xr, y_ref, ref, gy)
    if key in _GRU_REF:
        P, x, xr, y_ref, ref, gy = _GRU_REF[key]
    xd = x.reshape(T * B, I).to(device, BF16)
    wih = P["w_ih"].reshape(ndir * 3 * H, I).to(device, BF16)
    gi = _ops.gemm_nt(xd, wih, P["b_ih"].reshape(-1).to(device), gi_dtype)
    whh = P["w_hh"].to(device, BF16).contiguous()
    y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh, P["b_hh"].reshape(-1).to(device), T, B, H, ndir)
    gate("y", y.float().cpu().reshape(T, B, H), y_ref.detach())
    whhT = P["w_hh"].transpose(1, 2).contiguous().to(device, BF16)
    dbih = torch.zeros(ndir * 3 * H, device=device)
    dbhh = torch.zeros(ndir * 3 * H, device=device)
    dgi, dgh = _ops.gru_bwd(gy.reshape(T * B, H).to(device, BF16), gates, hseq, whhT, T, B, H, ndir, dbih, dbhh)
    dx = _ops.gemm_nt(dgi, wih.T.contiguous(), None, F32).cpu().reshape(T, B, I)
    gate("dx", dx, xr.grad)
    dwih = torch.zeros(ndir * 3 * H, I, device=device)
    _ops.gemm_tn_acc(dgi, xd, dwih)
    sufs = ["", "_reverse"][:ndir]
    ref_dwih = torch.cat([getattr(ref, "weight_ih_l0" + s).grad for s in sufs])
    gate("dwih", dwih.cpu(), ref_dwih)
    gate("dbih", dbih.cpu(), torch.cat([getattr(ref, "bias_ih_l0" + s).grad for s in sufs]))
    gate("dbhh", dbhh.cpu(), torch.cat([getattr(ref, "bias_hh_l0" + s).grad for s in sufs]))
    for d, s in enumerate(sufs):
        dwhh = torch.zeros(3 * H, H, device=device)
        a = dgh[:, d * 3 * H:(d + 1) * 3 * H]       # strided 2-d views, rows are (t, b)
        hb = hseq16[:, d * H:(d + 1) * H]
        if d == 0:      # h_{t-1} pairs with step t
            a, hb = a[B:], hb[:-B]
        else:           # reverse direction: h_{t+1} pairs with step t
            a, hb = a[:-B], hb[B:]
        _ops.gemm_tn_acc(a, hb, dwhh)
        gate("dwhh", dwhh.cpu(), getattr(ref, "weight_hh_l0" + s).grad)
    return errs


def test_clip_decay_adam(device):
    from asr import _ops
    rs = np.random.RandomState(0)
    n = 100003
    p = rs.randn(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    pd, md, vd = (torch.tensor(a).to(device) for a in (p, m, v))
    for step in (1, 2, 3):
        gr = (rs.randn(n) * (10.0 if step == 2 else 0.001)).astype(np.float32)     # step 2 is clipped, the others not
        gd = torch.tensor(gr).to(device)
        sq = torch.zeros(1, device=device)
        _ops.sqnorm_acc(gd, sq)
        np.testing.assert_allclose(sq.item(), (gr.astype(np.float64) ** 2).sum(), rtol=1e-5)
        _ops.clip_decay_adam(pd, gd, md, vd, 1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0, 1.0, sq, step)
        p, m, v = onn.clip_decay_adam(p.astype(np.float64), gr.astype(np.float64), m.astype(np.float64), v.astype(np.float64), step)
        np.testing.assert_allclose(pd.cpu().numpy(), p, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(md.cpu().numpy(), m, rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=1e-4, atol=1e-9)
        p, m, v = (a.astype(np.float32) for a in (p, m, v))


def test_cast_transpose_permute(device):
    from asr import _ops
    g = torch.Generator().manual_seed(2)
    w = torch.randn(70, 45, generator=g)
    assert torch.equal(_ops.cast_bf16(w.to(device)).cpu(), w.to(BF16))
    assert torch.equal(_ops.cast_bf16(w.to(device), transpose=True).cpu(), w.T.contiguous().to(BF16))
    x = torch.randn(3, 4, 5, 6, generator=g).to(device)
    y = _ops.permute4(x, (6, 3, 5, 4), (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), BF16)
    assert torch.equal(y.cpu(), x.permute(3, 0, 2, 1).contiguous().to(BF16).cpu())


def test_gru_abort_word_is_sticky_and_reported(device):
    """a set abort word survives later launches (they give up at once) and optimizer-side polling raises"""
    from asr import _ops, _lib
    T, B, H, ndir = 6, 4, 64, 2
    g = torch.Generator().manual_seed(0)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(device)
    whh = (torch.randn(ndir, 3 * H, H, generator=g) * 0.1).to(device).to(BF16)
    bhh = torch.zeros(ndir * 3 * H, device=device)
    _ops.gru_fwd(gi, whh, bhh, T, B, H, ndir)
    torch.cuda.synchronize()
    _ops.gru_check_sync()
    _ops.gru_poll_status(); torch.cuda.synchronize(); _ops.gru_poll_status()      # clean: no exception
    _ops.LAST_SYNC[0][1023:1024].fill_(1)                                          # pretend a launch gave up
    _ops.gru_fwd(gi, whh, bhh, T, B, H, ndir)                                     # must terminate (bounded, gives up at once)
    torch.cuda.synchronize()
    assert int(_ops.LAST_SYNC[0][1023:1024].cpu()[0]) != 0
    _ops.gru_poll_status()
    torch.cuda.synchronize()
    with pytest.raises(_lib.AsrHipError):
        _ops.gru_poll_status()
    torch.cuda.synchronize()
    y, *_ = _ops.gru_fwd(gi, whh, bhh, T, B, H, ndir)                             # the word was reset: back to normal
    torch.cuda.synchronize()
    _ops.gru_check_sync()
    assert torch.isfinite(y.float()).all()


def test_gru_full_size_forms_agree(device):
    """BASELINE size (T=1000, B=32, H=512, bidirectional): the persistent kernels (XCD-local hand-off, placement-free
    hand-off, forged split placement) against the one-launch-per-step kernels -- same arithmetic, so the states agree to
    bf16 rounding of the exchanged h, and the saved f32 states / gates to 1e-3."""
    from asr import _ops
    T, B, H, ndir = 1000, 32, 512, 2
    g = torch.Generator().manual_seed(1)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(device)
    whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(device)
    whh16 = whh.to(BF16).contiguous()
    whhT16 = whh.transpose(1, 2).contiguous().to(BF16)
    bhh = (torch.randn(ndir * 3 * H, generator=g) * 0.1).to(device)
    dy = (torch.randn(T * B, H, generator=g) * 0.1).to(device).to(BF16)
    res = {}
    try:
        for mode in (1, 2, 0, 4, 7, 10):
            _ops.GRU_MODE[0] = mode
            y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
            dbi = torch.zeros(ndir * 3 * H, device=device)
            dbh = torch.zeros(ndir * 3 * H, device=device)
            dgi, dgh = _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)
            torch.cuda.synchronize()
            _ops.gru_check_sync()
            # (the default kernel pair keeps its saved gates in IEEE half, blocked by workgroup: bring them to the standard layout)
            res[mode] = [t.float().clone() for t in (y, hseq, _ops.gru_gates_standard(gates, H), dgi, dgh, dbi, dbh)]
            res[mode].append(gates.dtype)
    finally:
        _ops.GRU_MODE[0] = 0
    ref = res[1]
    for mode in (2, 0, 4, 7, 10):
        for name, a, r, tol in zip(("y", "hseq", "gates", "dgi", "dgh", "db_ih", "db_hh"), res[mode], ref,
                                   (2e-2, 2e-3, 2e-3, 2e-2, 2e-2, 1e-2, 1e-2)):
            err = float((a - r).abs().max()) / (float(r.abs().max()) + 1e-12)
            assert err < tol, (mode, name, err)
    # the XCD-local form and its forged-placement fall-back run the same instruction sequence per workgroup (only the
    # hand-off differs): bit-identical (bias gradients excepted: their float atomics land in any order).  Mode 2 runs the
    # wide forward kernel (another K split), so it agrees to rounding only.
    # (mode 0 = payload polled as its own signal, mode 4 = flag line, mode 7 = counters after a forged split placement)
    # (backward: mode 0 runs the partial-sum exchange kernel, mode 10 the same with a forged split placement; modes 4 / 7 the
    # wide kernel and its forged fall-back)
    for other in (4, 7, 10):
        for a, b_ in zip(res[0][:2], res[other][:2]):
            assert torch.equal(a, b_)
        a, b_ = res[0][2], res[other][2]            # saved gates: the same values, rounded to half where the pair keeps them so
        if res[0][7] != res[other][7]:
            a, b_ = a.to(torch.float16), b_.to(torch.float16)
        assert torch.equal(a, b_)
    for a, b_ in zip(res[0][3:5], res[10][3:5]):
        assert torch.equal(a, b_)
    for a, b_ in zip(res[4][3:5], res[7][3:5]):
        assert torch.equal(a, b_)

def test_non_finite_step_is_skipped(device):
    """a NaN / Inf gradient norm leaves parameters and optimiser state untouched (run/ctc/cnn/train.py:193-197)"""
    from asr import _ops
    n = 4097
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(n, generator=g).to(device)
    for bad in (float("nan"), float("inf")):
        p, m, v = p0.clone(), torch.full((n,), 0.25, device=device), torch.full((n,), 0.5, device=device)
        gr = torch.randn(n, generator=g).to(device)
        gr[17] = bad
        sq = torch.zeros(1, device=device)
        _ops.sqnorm_acc(gr, sq)
        _ops.clip_decay_adam(p, gr, m, v, 1e-3, 0.9, 0.999, 1e-8, 1e-5, 1.0, 1.0, sq, 1)
        assert torch.equal(p, p0) and bool((m == 0.25).all()) and bool((v == 0.5).all())
        vel = torch.zeros(n, device=device)
        _ops.clip_decay_sgd(p, gr, vel, 1, 1e-2, 0.9, 0.0, 0.0, 1.0, sq)
        assert torch.equal(p, p0) and bool((vel == 0).all())


def test_step_control_with_a_loss_scale_on_the_device(device):
    """asr_step_control_scaled (chainer.Optimizer.loss_scaling): the gradient factor carries 1 / S, a non-finite norm that no recurrence
    explains halves S and drops the step, `interval` applied steps in a row double it, a static scale (interval 0) never moves, a step
    dropped because a recurrence gave up leaves S alone -- all on the device, read back here"""
    from asr import _ops
    n = 1000
    g = torch.Generator().manual_seed(3)
    grad = torch.randn(n, generator=g).to(device)
    partials = torch.empty(_ops.sqnorm_partials_count(n), device=device)
    applied = torch.zeros(1, dtype=torch.int32, device=device)
    ctl = torch.zeros(8, device=device)
    no_abort = torch.zeros(1, dtype=torch.int32, device=device)
    ls = torch.tensor([1024.0, 0.0, 3.0, 0.0], device=device)          # S, applied since the last change, interval, overflows
    norm = float(grad.norm())

    def step(gr, abort=no_abort, scale=ls):
        _ops.step_control(gr, partials, 1.0, 0.5, 1e-3, 0.9, 0.999, applied, ctl, abort, -1, scale)
        return ctl.cpu().tolist(), (None if scale is None else scale.cpu().tolist()), int(applied.cpu()[0])

    c, s, a = step(grad * 1024.0)                                       # a scaled gradient: factor = 0.5 / 1024, clipped by the UNSCALED norm
    assert c[0] == 0.0 and a == 1 and s == [1024.0, 1.0, 3.0, 0.0]
    want = 0.5 / 1024.0 * min(1.0, 1.0 / (norm * 0.5))
    assert abs(c[1] - want) < 1e-6 * want, (c[1], want)
    c, s, a = step(grad * 1024.0); c, s, a = step(grad * 1024.0)        # three applied steps in a row: S doubles
    assert a == 3 and s == [2048.0, 0.0, 3.0, 0.0]
    bad = grad.clone(); bad[5] = float("inf")
    c, s, a = step(bad)                                                 # overflow: dropped, S halved, counted
    assert c[0] == 1.0 and c[5] == 0.0 and a == 3 and s == [1024.0, 0.0, 3.0, 1.0]
    raised = torch.ones(1, dtype=torch.int32, device=device)
    c, s, a = step(grad, abort=raised)                                  # a given-up recurrence: dropped, S untouched
    assert c[0] == 1.0 and c[5] == 1.0 and a == 3 and s == [1024.0, 0.0, 3.0, 1.0]
    static = torch.tensor([256.0, 0.0, 0.0, 0.0], device=device)
    for _ in range(5):
        c, s, a = step(grad * 256.0, scale=static)
    assert s[0] == 256.0 and a == 8
    c, s, a = step(bad, scale=static)                                   # a static scale stays put on an overflow too: dropped, counted
    assert c[0] == 1.0 and a == 8 and s == [256.0, 0.0, 0.0, 1.0]
    tiny = torch.tensor([1.0, 0.0, 7.0, 0.0], device=device)            # a dynamic S may go below 1 (gradients beyond the half range WITHOUT a scale)
    c, s, a = step(bad, scale=tiny)
    assert s[0] == 0.5 and s[3] == 1.0
    c, s, a = step(grad, scale=None)                                    # no scale: asr_step_control
    assert abs(c[1] - 0.5 * min(1.0, 1.0 / (norm * 0.5))) < 1e-6 and a == 9


@pytest.mark.parametrize("T,B,Hin,Ci,Co,KH,KW,ph,causal", [(23, 3, 13, 64, 128, 3, 5, 0, True), (17, 2, 6, 32, 96, 3, 5, 1, True),
                                                          (9, 2, 7, 64, 64, 1, 3, 0, False), (300, 4, 13, 64, 128, 3, 5, 0, True),
                                                          # > 128 output (or, for backward-data, input) channels: 256 x 256 tiles
                                                          (23, 3, 13, 64, 256, 3, 5, 1, True), (11, 2, 7, 128, 192, 3, 5, 1, True),
                                                          (12, 2, 6, 256, 64, 3, 5, 1, True),
                                                          # >= 400 tiles of 256 rows: the persistent kernels (128- and 64-wide)
                                                          (600, 16, 13, 64, 128, 3, 5, 0, True), (600, 16, 13, 32, 64, 3, 5, 1, True),
                                                          # 8 input channels (a padded first layer): one tap per 16-B chunk
                                                          (600, 16, 13, 8, 128, 3, 5, 0, True), (500, 16, 20, 8, 64, 3, 3, 1, False)])
def test_implicit_conv_matches_im2col_gemm(device, T, B, Hin, Ci, Co, KH, KW, ph, causal):
    """asr_conv_nt (no column matrix) against the im2col + GEMM path it replaces, forward and backward-data: the same
    products in the same bf16 operands, so they agree to accumulation order"""
    from asr import _ops
    rs = np.random.RandomState(T + Ci)
    pt = KW - 1
    Tout = T if causal else T + 2 * pt - KW + 1
    Hout = Hin + 2 * ph - KH + 1
    x = torch.from_numpy(rs.randn(T, B, Hin, Ci).astype(np.float32)).to(device).to(BF16)
    W = torch.from_numpy((rs.randn(Co, Ci, KH, KW) * 0.1).astype(np.float32)).to(device)
    bias = torch.from_numpy(rs.randn(Co).astype(np.float32)).to(device)
    w16 = _ops.conv_weight_pack(W)
    col = _ops.im2col(x, (x.stride(0), x.stride(1), x.stride(2), x.stride(3)), T, B, Hin, Ci, KH, KW, ph, pt, Tout)
    ref = _ops.gemm_nt(col, w16, bias, torch.float32)
    w16c = w16 if (KH * KW * Ci) % 32 == 0 else _ops.conv_weight_pack(W, Kp=(KH * KW * Ci + 31) // 32 * 32)      # K step of the kernels
    got = _ops.conv_nt(x, w16c, bias, torch.float32, KH, KW, ph, pt, +1, Tout, Hout)
    assert _rel(got.cpu(), ref.cpu()) < 1e-5
    got16 = _ops.conv_nt(x, w16c, bias, BF16, KH, KW, ph, pt, +1, Tout, Hout)
    assert _rel(got16.float().cpu(), ref.cpu()) < 1e-2
    # backward-data: dx = col2im(gy . W)
    gy = torch.from_numpy(rs.randn(Tout, B, Hout, Co).astype(np.float32)).to(device).to(BF16)
    w16t = _ops.conv_weight_pack(W, transpose=True)
    dcol = _ops.gemm_nt(gy.reshape(-1, Co), w16t, None, torch.float32)
    ref_dx = _ops.col2im(dcol.to(BF16), T, B, Hin, Ci, KH, KW, ph, pt, Tout).float()
    wb = _ops.conv_weight_pack_bwd(W)
    got_dx = _ops.conv_nt(gy, wb, None, torch.float32, KH, KW, ph, pt, -1, T, Hin).reshape(T, B, Hin, Ci)
    assert _rel(got_dx.cpu(), ref_dx.cpu()) < 2e-2      # the reference path rounds dcol to bf16 before the gather
    # weight gradient: gy^T . col without the column matrix (asr_conv_tn_acc), accumulated on top of what is there
    K = KH * KW * Ci
    ref_w = torch.ones(Co, col.shape[1], device=device)
    _ops.gemm_tn_acc(gy.reshape(-1, Co), col, ref_w)
    got_w = torch.ones(Co, K, device=device)
    _ops.conv_tn_acc(gy.reshape(-1, Co), x, got_w, KH, KW, ph, pt, Tout, Hout)
    assert _rel(got_w.cpu(), ref_w[:, :K].cpu()) < 1e-5
    gW_ref, gW = torch.zeros_like(W), torch.zeros_like(W)
    _ops.conv_weight_grad_unpack(ref_w, gW_ref)
    _ops.conv_weight_grad_unpack(got_w, gW, Ci)
    assert _rel(gW.cpu(), gW_ref.cpu()) < 1e-5


@pytest.mark.parametrize("T,B,Hin,Ci,Co,KH,KW,ph,causal", [(23, 3, 13, 64, 256, 3, 5, 1, True), (11, 2, 7, 128, 192, 3, 5, 1, True), (12, 2, 6, 256, 64, 3, 5, 1, True),
                                                          (9, 2, 7, 64, 68, 1, 3, 0, False), (200, 8, 13, 256, 512, 3, 5, 1, True), (77, 2, 1, 64, 72, 1, 1, 0, True)])
def test_conv_nt_8ph_matches_im2col_gemm(device, T, B, Hin, Ci, Co, KH, KW, ph, causal):
    """asr_conv_nt_8ph (csrc/gemm8.hip: the implicit convolution on the eight-wave kernel, tap walk as scalar state per operand half)
    against im2col + GEMM, forward and backward-data, float32 and bf16 outputs; causal and symmetric time padding, a 1 x 1 kernel"""
    from asr import _ops
    rs = np.random.RandomState(T + Ci)
    pt = KW - 1
    Tout = T if causal else T + 2 * pt - KW + 1
    Hout = Hin + 2 * ph - KH + 1
    x = torch.from_numpy(rs.randn(T, B, Hin, Ci).astype(np.float32)).to(device).to(BF16)
    W = torch.from_numpy((rs.randn(Co, Ci, KH, KW) * 0.1).astype(np.float32)).to(device)
    bias = torch.from_numpy(rs.randn(Co).astype(np.float32)).to(device)
    w16 = _ops.conv_weight_pack(W)
    col = _ops.im2col(x, (x.stride(0), x.stride(1), x.stride(2), x.stride(3)), T, B, Hin, Ci, KH, KW, ph, pt, Tout)
    ref = _ops.gemm_nt(col, w16, bias, torch.float32)
    K = KH * KW * Ci
    w16c = w16 if K % 64 == 0 else _ops.conv_weight_pack(W, Kp=(K + 63) // 64 * 64)      # (empty taps behind: the K step of the kernel)
    got = _ops.conv_nt_8ph(x, w16c, bias, torch.float32, KH, KW, ph, pt, +1, Tout, Hout)
    assert _rel(got.cpu(), ref.cpu()) < 1e-5
    got16 = _ops.conv_nt_8ph(x, w16c, bias, BF16, KH, KW, ph, pt, +1, Tout, Hout)
    assert _rel(got16.float().cpu(), ref.cpu()) < 1e-2
    if Co % 64 == 0:                        # backward-data: the gradient's channels are the K side
        gy = torch.from_numpy(rs.randn(Tout, B, Hout, Co).astype(np.float32)).to(device).to(BF16)
        wb = _ops.conv_weight_pack_bwd(W)
        if wb.shape[1] % 64 == 0 and Ci % 4 == 0:
            ref_dx = _ops.conv_nt(gy, wb, None, torch.float32, KH, KW, ph, pt, -1, T, Hin)
            got_dx = _ops.conv_nt_8ph(gy, wb, None, torch.float32, KH, KW, ph, pt, -1, T, Hin)
            assert _rel(got_dx.cpu(), ref_dx.cpu()) < 1e-5


@pytest.mark.parametrize("T,B,Hin,Ci,Co,KH,KW,ph,causal", [(23, 3, 13, 64, 128, 3, 5, 0, True), (41, 3, 11, 128, 64, 3, 5, 0, True), (9, 2, 7, 64, 64, 1, 3, 0, False),
                                                          (300, 4, 13, 64, 128, 3, 5, 1, True), (77, 2, 1, 64, 72, 1, 1, 0, True), (150, 4, 13, 128, 40, 3, 5, 1, True),
                                                          (33, 2, 5, 192, 100, 3, 3, 1, False)])
def test_conv_nt_8pn_matches_im2col_gemm(device, T, B, Hin, Ci, Co, KH, KW, ph, causal):
    """asr_conv_nt_8pn (csrc/gemm8.hip: the narrow tiles, 256 x 64 with a four-stage ring that takes all of LDS, 256 x 128 with three
    stages) against im2col + GEMM, forward (N = Co <= 128) and -- where the gradient's channels are a multiple of 64 -- backward-data
    (N = Ci), float32 and bf16 outputs, ragged N"""
    from asr import _ops
    rs = np.random.RandomState(T + Ci + 1)
    pt = KW - 1
    Tout = T if causal else T + 2 * pt - KW + 1
    Hout = Hin + 2 * ph - KH + 1
    x = torch.from_numpy(rs.randn(T, B, Hin, Ci).astype(np.float32)).to(device).to(BF16)
    W = torch.from_numpy((rs.randn(Co, Ci, KH, KW) * 0.1).astype(np.float32)).to(device)
    bias = torch.from_numpy(rs.randn(Co).astype(np.float32)).to(device)
    w16 = _ops.conv_weight_pack(W)
    col = _ops.im2col(x, (x.stride(0), x.stride(1), x.stride(2), x.stride(3)), T, B, Hin, Ci, KH, KW, ph, pt, Tout)
    ref = _ops.gemm_nt(col, w16, bias, torch.float32)
    K = KH * KW * Ci
    w16c = w16 if K % 64 == 0 else _ops.conv_weight_pack(W, Kp=(K + 63) // 64 * 64)
    got = _ops.conv_nt_8pn(x, w16c, bias, torch.float32, KH, KW, ph, pt, +1, Tout, Hout)
    assert _rel(got.cpu(), ref.cpu()) < 1e-5
    got16 = _ops.conv_nt_8pn(x, w16c, bias, BF16, KH, KW, ph, pt, +1, Tout, Hout)
    assert _rel(got16.float().cpu(), ref.cpu()) < 1e-2
    if Co % 64 == 0 and Ci <= 128 and Ci % 4 == 0:
        gy = torch.from_numpy(rs.randn(Tout, B, Hout, Co).astype(np.float32)).to(device).to(BF16)
        wb = _ops.conv_weight_pack_bwd(W)
        if wb.shape[1] % 64 == 0:
            dcol = _ops.gemm_nt(gy.reshape(-1, Co), _ops.conv_weight_pack(W, transpose=True), None, torch.float32)
            ref_dx = _ops.col2im(dcol.to(BF16), T, B, Hin, Ci, KH, KW, ph, pt, Tout).float().reshape(-1, Ci)
            got_dx = _ops.conv_nt_8pn(gy, wb, None, torch.float32, KH, KW, ph, pt, -1, T, Hin)
            assert _rel(got_dx.cpu(), ref_dx.cpu()) < 2e-2      # the reference path rounds dcol to bf16 before the gather


@pytest.mark.parametrize("T,B,Hin,Ci,Co,KH,KW,ph,causal", [(23, 3, 13, 64, 256, 3, 5, 1, True), (17, 2, 6, 32, 96, 3, 5, 1, True), (9, 2, 7, 64, 72, 1, 3, 0, False),
                                                          (200, 8, 13, 128, 256, 3, 5, 1, True), (77, 2, 1, 64, 72, 1, 1, 0, True), (120, 5, 38, 8, 64, 3, 5, 0, True)])
def test_conv_tn_8ph_matches_im2col_gemm(device, T, B, Hin, Ci, Co, KH, KW, ph, causal):
    """asr_conv_tn_acc_8ph (csrc/gemm8.hip: the weight gradient without a column matrix, the im2col rows walked incrementally per loader
    slot) against gemm_tn_acc on the materialised column matrix -- small-integer operands give exact sums --, accumulated on top of what
    is there; padded taps, a 1 x 1 kernel, 8 input channels (one tap per 16-byte chunk)"""
    from asr import _ops
    rs = np.random.RandomState(T + Ci + Co)
    pt = KW - 1
    Tout = T if causal else T + 2 * pt - KW + 1
    Hout = Hin + 2 * ph - KH + 1
    x = torch.from_numpy(rs.randint(-3, 4, size=(T, B, Hin, Ci)).astype(np.float32)).to(device).to(BF16)
    gy = torch.from_numpy(rs.randint(-3, 4, size=(Tout, B, Hout, Co)).astype(np.float32)).to(device).to(BF16)
    col = _ops.im2col(x, (x.stride(0), x.stride(1), x.stride(2), x.stride(3)), T, B, Hin, Ci, KH, KW, ph, pt, Tout)
    K = KH * KW * Ci
    ref = (gy.reshape(-1, Co).float().T @ col.float())[:, :K] + 2.0
    got = torch.full((Co, K), 2.0, device=device)
    _ops.conv_tn_acc_8ph(gy.reshape(-1, Co), x, got, KH, KW, ph, pt, Tout, Hout)
    assert torch.equal(got.cpu(), ref.cpu())


@pytest.mark.parametrize("T,B,Hin,Ci,Co,KH,KW,ph,causal", [(23, 3, 13, 64, 128, 3, 5, 0, True), (17, 2, 6, 32, 96, 3, 5, 1, True),
                                                          (9, 2, 7, 64, 64, 1, 3, 0, False), (300, 4, 13, 64, 128, 3, 5, 0, True),
                                                          (23, 3, 13, 64, 256, 3, 5, 1, True), (11, 2, 7, 128, 192, 3, 5, 1, True),
                                                          (12, 2, 6, 256, 64, 3, 5, 1, True), (120, 5, 38, 32, 64, 3, 5, 0, True),
                                                          (41, 3, 11, 128, 64, 3, 5, 0, True), (77, 2, 1, 64, 72, 1, 1, 0, True)])
def test_direct_conv_equals_the_implicit_gemm_conv(device, T, B, Hin, Ci, Co, KH, KW, ph, causal):
    """asr_conv_direct_nt (activation block resident in LDS, csrc/conv_direct.hip) against the implicit-GEMM kernels of asr_conv_nt on the
    same bf16 operands, forward (with bias) and backward-data: the same products summed in the same K order by the same MFMA -- equal up
    to the bf16 rounding of the output (the float32 sums can differ in the last bit only where the MFMA's internal order does not)"""
    from asr import _lib, _ops
    rs = np.random.RandomState(T + Ci + Co)
    pt = KW - 1
    Tout = T if causal else T + 2 * pt - KW + 1
    Hout = Hin + 2 * ph - KH + 1
    x = torch.from_numpy(rs.randn(T, B, Hin, Ci).astype(np.float32)).to(device).to(BF16)
    W = torch.from_numpy((rs.randn(Co, Ci, KH, KW) * 0.1).astype(np.float32)).to(device)
    bias = torch.from_numpy(rs.randn(Co).astype(np.float32)).to(device)
    K = KH * KW * Ci
    w16 = _ops.conv_weight_pack(W, Kp=(K + 31) // 32 * 32)
    assert _lib.lib().asr_conv_direct_ok(T, B, Hin, Ci, KH, KW, Tout, Hout, Co, w16.shape[1], 1) == 1
    import os
    ref = _conv_nt_implicit(x, w16, bias, KH, KW, ph, pt, +1, Tout, Hout)
    got = _ops.conv_direct_nt(x, w16, bias, KH, KW, ph, pt, +1, Tout, Hout)
    assert got.shape == ref.shape and _rel(got.float().cpu(), ref.float().cpu()) < 2e-3
    assert float((got.float() - ref.float()).abs().max()) <= 2.0 ** -6 * float(ref.float().abs().max())        # at most one bf16 ulp anywhere
    if Co in (32, 64, 128, 256):        # backward-data: the roles of the channels swap
        gy = torch.from_numpy(rs.randn(Tout, B, Hout, Co).astype(np.float32)).to(device).to(BF16)
        wb = _ops.conv_weight_pack_bwd(W)
        if _lib.lib().asr_conv_direct_ok(Tout, B, Hout, Co, KH, KW, T, Hin, Ci, wb.shape[1], 1) != 1:
            assert Co == 256 and Hout >= 13         # (the block of 256 channels x 15 heights x 23 time steps does not fit the LDS)
            return
        ref_dx = _conv_nt_implicit(gy, wb, None, KH, KW, ph, pt, -1, T, Hin)
        got_dx = _ops.conv_direct_nt(gy, wb, None, KH, KW, ph, pt, -1, T, Hin)
        assert _rel(got_dx.float().cpu(), ref_dx.float().cpu()) < 2e-3
        assert float((got_dx.float() - ref_dx.float()).abs().max()) <= 2.0 ** -6 * float(ref_dx.float().abs().max())


@pytest.mark.parametrize("T,B,Hin,Ci,Co,KH,KW,ph,causal,k", [(50, 3, 40, 3, 128, 3, 5, 0, True, 3), (37, 2, 17, 1, 256, 3, 3, 1, False, 2),
                                                            (29, 5, 11, 3, 128, 2, 4, 0, True, 4), (9, 1, 40, 3, 128, 3, 5, 0, True, 3),
                                                            (300, 8, 40, 3, 128, 3, 5, 0, True, 3), (41, 3, 7, 7, 128, 5, 3, 2, True, 3),
                                                            (3, 1, 4, 3, 128, 3, 5, 0, True, 3),           # two output heights: ONE short window
                                                            (1, 2, 9, 2, 128, 3, 1, 0, True, 2),           # one frame per utterance, a (3, 1) filter
                                                            (2, 1, 40, 3, 128, 3, 5, 0, False, 3)])        # fewer frames than filter taps in time
def test_fused_first_block_equals_convolution_maxout_pooling(device, T, B, Hin, Ci, Co, KH, KW, ph, causal, k):
    """csrc/conv_first.hip against the passes it replaces on the same operands -- asr_conv_nt (bf16 out) -> asr_maxout2_pool_fwd, and
    asr_maxout2_pool_bwd_db -> asr_conv_tn_acc -> unpack: the pooled values bit for bit (the same products in the same K order through the
    same MFMA; one bf16 ulp allowed where its internal order differs), the winners' indices consistent with them, the weight and bias
    gradients to float32 summation order"""
    from asr import _ops
    rs = np.random.RandomState(T + Hin + Co + k)
    pt = KW - 1
    Tout = T if causal else T + 2 * pt - KW + 1
    Hout = Hin + 2 * ph - KH + 1
    assert _ops.conv_mp_ok(Ci, KH, KW, Co, k)
    x = torch.from_numpy(rs.randn(B, Ci, Hin, T).astype(np.float32)).to(device)
    W = torch.from_numpy((rs.randn(Co, Ci, KH, KW) * 0.3).astype(np.float32)).to(device)
    bias = torch.from_numpy(rs.randn(Co).astype(np.float32)).to(device)
    x8 = _ops.pack_input_pad(x, (x.stride(3), x.stride(0), x.stride(2), x.stride(1)), T, B, Hin, Ci, 8)
    Wp = torch.zeros((Co, 8, KH, KW), device=device)
    Wp[:, :Ci] = W
    w128 = _ops.conv_weight_pack(Wp, Kp=128)
    # the three passes
    conv = _ops.conv_nt(x8, w128, bias, BF16, KH, KW, ph, pt, +1, Tout, Hout).reshape(Tout, B, Hout, Co)
    want = _ops.maxout2_pool_fwd(conv, k)
    got, idx = _ops.conv_mp_fwd(x8, w128, bias, KH, KW, ph, pt, Tout, Hout, k)
    torch.cuda.synchronize()
    assert got.shape == want.shape and idx.shape == want.shape
    gf, wf = got.float().cpu(), want.float().cpu()
    assert float((gf - wf).abs().max()) <= 2.0 ** -6 * float(wf.abs().max())
    assert float((gf != wf).float().mean()) < 1e-3
    # the index names an element of the window that holds the pooled value
    Hp = want.shape[2]
    cpad = torch.full((Tout, B, Hp * k, Co), float("-inf"))
    cpad[:, :, :Hout] = conv.float().cpu()
    cand = cpad.reshape(Tout, B, Hp, k, Co // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(Tout, B, Hp, Co // 2, 2 * k)
    ix = idx.cpu().long()
    assert int(ix.max()) < 2 * k
    picked = torch.gather(cand, 4, ix.unsqueeze(-1)).squeeze(-1)
    assert float((picked - gf).abs().max()) <= 2.0 ** -6 * float(wf.abs().max())
    same = gf == wf
    first = cand.argmax(dim=4)          # first maximum in (row, channel) order = the tie rule of asr_maxout2_pool_bwd
    assert float((ix[same] != first[same]).float().mean()) < 1e-3
    # backward
    gy = torch.from_numpy(rs.randn(Tout, B, Hp, Co // 2).astype(np.float32)).to(device).to(BF16)
    gW_ref = torch.zeros(Co, Ci, KH, KW, device=device)
    gb_ref = torch.zeros(Co, device=device)
    g = _ops.maxout2_pool_bwd(conv, gy, k)
    scratch = torch.zeros(Co, KH * KW * 8, device=device)
    _ops.conv_tn_acc(g.reshape(Tout * B * Hout, Co), x8, scratch, KH, KW, ph, pt, Tout, Hout)
    _ops.conv_weight_grad_unpack(scratch, gW_ref, 8)
    _ops.colsum_acc(g.reshape(Tout * B * Hout, Co), gb_ref)
    gW = torch.full((Co, Ci, KH, KW), 0.5, device=device)          # the entry ACCUMULATES
    gb = torch.full((Co,), -0.25, device=device)
    _ops.conv_mp_bwd(gy, idx, x8, gW, gb, KH, KW, ph, pt, Hout, k)
    torch.cuda.synchronize()
    # (where an ulp-different convolution value changed a winner the two gradients route one element differently: the bars allow for it)
    assert _rel((gW - 0.5).cpu(), gW_ref.cpu()) < 2e-3, _rel((gW - 0.5).cpu(), gW_ref.cpu())
    assert _rel((gb + 0.25).cpu(), gb_ref.cpu()) < 2e-3
    gW2 = torch.zeros(Co, Ci, KH, KW, device=device)
    _ops.conv_mp_bwd(gy, idx, x8, gW2, None, KH, KW, ph, pt, Hout, k)
    torch.cuda.synchronize()
    assert _rel(gW2.cpu(), (gW - 0.5).cpu()) < 1e-5


def test_fused_first_block_backward_general_form_in_a_child_process(device):
    """the backward kernel has two forms -- one frame per iteration (at most 16 pooling windows per frame: the default wherever it applies) and
    the general one (virtual im2col rows); ASR_DEBUG conv_mp_frame=0 selects the general form for every shape: the same test cases again"""
    import os, subprocess, sys
    if "conv_mp_frame=0" in os.environ.get("ASR_DEBUG", ""):
        pytest.skip("this IS the child process")
    env = dict(os.environ, ASR_DEBUG=",".join(filter(None, [os.environ.get("ASR_DEBUG", ""), "conv_mp_frame=0"])))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "tests/test_kernels_gpu.py", "-k",
                        "fused_first_block_equals"], cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-4000:]


def _conv_nt_implicit(x, W2, bias, KH, KW, pad_h, pad_t, sgn, Tr, Hr):
    """asr_conv_nt with the direct kernel switched off: the float32-output form never dispatches to it"""
    from asr import _ops
    return _ops.conv_nt(x, W2, bias, torch.float32, KH, KW, pad_h, pad_t, sgn, Tr, Hr).to(BF16)


@pytest.mark.parametrize("B,C,H,T", [(3, 3, 40, 130), (2, 1, 13, 64), (2, 8, 9, 65), (32, 3, 40, 1000)])
def test_pack_input_pad_layouts(device, B, C, H, T):
    """(T, B, H, 8) bf16 rows with zero channels behind C, from the loader's (B, C, H, T) float32 (time innermost: the LDS-tile kernel)
    and from a time-outermost copy of it (the general gather): both must equal the rounded source exactly"""
    from asr import _ops
    g = torch.Generator().manual_seed(T + H)
    x = torch.randn(B, C, H, T, generator=g)
    want = torch.zeros(T, B, H, 8)
    want[..., :C] = x.permute(3, 0, 2, 1).to(BF16).float()
    xd = x.to(device)
    got = _ops.pack_input_pad(xd, (xd.stride(3), xd.stride(0), xd.stride(2), xd.stride(1)), T, B, H, C, 8)
    assert xd.stride(3) == 1 and torch.equal(got.float().cpu().reshape(T, B, H, 8), want)
    xt = xd.permute(3, 0, 2, 1).contiguous()                  # (T, B, H, C): channels innermost
    got2 = _ops.pack_input_pad(xt, (xt.stride(0), xt.stride(1), xt.stride(2), xt.stride(3)), T, B, H, C, 8)
    assert torch.equal(got2.float().cpu().reshape(T, B, H, 8), want)


def test_first_layer_weight_gradient_with_padded_channels(device):
    """first conv layer: 3 input channels zero-padded to 8 for the implicit kernels; the weight gradient comes back
    through a scratch with channel pitch 8 (asr_conv_weight_grad_unpack Cs = 8) and must equal the float64 correlation"""
    from asr import _ops
    rs = np.random.RandomState(4)
    T, B, Hin, Ci, Co, KH, KW, ph = 21, 2, 9, 3, 32, 3, 5, 1
    pt, Tout, Hout = KW - 1, T, Hin + 2 * ph - KH + 1
    x = torch.from_numpy(rs.randn(B, Ci, Hin, T).astype(np.float32)).to(BF16).float()
    gy = torch.from_numpy(rs.randn(Tout, B, Hout, Co).astype(np.float32)).to(BF16).float()
    xd = x.to(device)
    xpad = _ops.pack_input_pad(xd, (xd.stride(3), xd.stride(0), xd.stride(2), xd.stride(1)), T, B, Hin, Ci, 8)
    scratch = torch.zeros(Co, KH * KW * 8, device=device)
    _ops.conv_tn_acc(gy.to(device).to(BF16).reshape(-1, Co), xpad, scratch, KH, KW, ph, pt, Tout, Hout)
    gW = torch.zeros(Co, Ci, KH, KW, device=device)
    _ops.conv_weight_grad_unpack(scratch, gW, 8)
    xp = torch.nn.functional.pad(x.double(), (pt, 0, ph, ph))                      # (B, Ci, Hin + 2 ph, T + pt)
    ref = torch.zeros(Co, Ci, KH, KW, dtype=torch.float64)
    g = gy.double().permute(1, 3, 2, 0)                                            # (B, Co, Hout, Tout)
    for kh in range(KH):
        for kw in range(KW):
            ref[:, :, kh, kw] = torch.einsum("bohs,bchs->oc", g, xp[:, :, kh:kh + Hout, kw:kw + Tout])
    assert _rel(gW.cpu().double(), ref) < 1e-5
    # one scratch copy per XCD (a single output tile under hundreds of K splits): the same gradient after the unpack's sum
    assert _ops.conv_tn_copies(Co, 8, KH, KW) == 8 and _ops.conv_tn_copies(512, 256, KH, KW) == 1
    scratch8 = torch.zeros(8, Co, KH * KW * 8, device=device)
    _ops.conv_tn_acc(gy.to(device).to(BF16).reshape(-1, Co), xpad, scratch8, KH, KW, ph, pt, Tout, Hout)
    gW8 = torch.ones(Co, Ci, KH, KW, device=device)
    _ops.conv_weight_grad_unpack(scratch8, gW8, 8)
    assert _rel((gW8 - 1.0).cpu().double(), ref) < 1e-5


def test_persistent_recurrence_waits_out_busy_cus(device):
    """a persistent launch that cannot be fully resident at once -- half of the CUs are held for 30 ms by
    LDS-heavy workgroups of another stream, the way a resident collective or GEMM would hold them -- must simply start late:
    the workgroups that did get a CU poll (bounded, seconds) until the others arrive, no abort, same results."""
    from asr import _ops, _lib
    T, B, H, ndir = 300, 32, 512, 2
    g = torch.Generator().manual_seed(3)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(device)
    whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(device)
    whh16, whhT16 = whh.to(BF16).contiguous(), whh.transpose(1, 2).contiguous().to(BF16)
    bhh = (torch.randn(ndir * 3 * H, generator=g) * 0.1).to(device)
    dy = (torch.randn(T * B, H, generator=g) * 0.1).to(device).to(BF16)

    def run(hog):
        side = torch.cuda.Stream()
        torch.cuda.synchronize()
        if hog:
            _lib.check(_lib.lib().asr_occupy_cus(side.cuda_stream, 30000, 120 * 1024, 128), "asr_occupy_cus")
        y, hseq, hseq16, gates = _ops.gru_fwd(gi, whh16, bhh, T, B, H, ndir)
        if hog:
            _lib.check(_lib.lib().asr_occupy_cus(side.cuda_stream, 30000, 120 * 1024, 128), "asr_occupy_cus")
        dgi, dgh = _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, None, None)
        torch.cuda.synchronize()
        _ops.gru_check_sync()
        return [t.clone() for t in (y, hseq, gates, dgi, dgh)]

    ref = run(False)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    got = run(True)
    t1.record()
    torch.cuda.synchronize()
    for a, b_ in zip(ref, got):
        assert torch.equal(a, b_)
    assert t0.elapsed_time(t1) > 25.0          # the hog really was in the way


def test_persistent_recurrence_beside_a_resident_collective_stand_in(device):
    """a collective kernel that is RESIDENT while a recurrence runs -- 16 workgroups that stream memory with 16 KB of LDS
    each (asr_stream_traffic), i.e. small enough to sit beside a recurrence workgroup on its CU -- costs the recurrence time (measured
    +14 .. +35 %, tools/gru_beside_collective.py) but neither an abort nor a different result: the hand-off protocol depends on no timing."""
    from asr import _ops, _lib
    T, B, H, ndir = 400, 32, 512, 2
    g = torch.Generator().manual_seed(5)
    gi = torch.randn(T * B, ndir * 3 * H, generator=g).to(device).to(_ops.gru_gi_dtype(T, B, H, ndir))
    whh = (torch.randn(ndir, 3 * H, H, generator=g) / H ** 0.5).to(device)
    whh16, whhT16 = whh.to(BF16).contiguous(), whh.transpose(1, 2).contiguous().to(BF16)
    bhh = (torch.randn(ndir * 3 * H, generator=g) * 0.1).to(device)
    dy = (torch.randn(T * B, H, generator=g) * 0.1).to(device).to(BF16)
    scratch = torch.zeros(16 * 1024 * 1024, device=device)
    side = torch.cuda.Stream()

    def run(beside):
        torch.cuda.synchronize()
        if beside:
            _lib.check(_lib.lib().asr_stream_traffic(side.cuda_stream, 20000, 16 * 1024, 16, scratch.data_ptr(), scratch.numel() * 4), "asr_stream_traffic")
            _lib.lib().asr_stream_delay(_lib.stream(), 300)
        y, hseq, hseq16, gates = _ops.gru_fwd(gi.clone(), whh16, bhh, T, B, H, ndir)
        dbi, dbh = torch.zeros(ndir * 3 * H, device=device), torch.zeros(ndir * 3 * H, device=device)
        dgi, dgh = _ops.gru_bwd(dy, gates, hseq, whhT16, T, B, H, ndir, dbi, dbh)
        torch.cuda.synchronize()
        _ops.gru_check_sync()
        return [t.clone() for t in (y, hseq, _ops.gru_gates_standard(gates, H), dgi, dgh)]
    ref, got = run(False), run(True)
    for a, b_ in zip(ref, got):
        assert torch.equal(a, b_)


@pytest.mark.parametrize("mode", ["1", "2"])
def test_wide_nt_kernels_in_a_forced_process(mode):
    """the 256 x 256 NT kernels (csrc/gemm.hip gemm_nt_wide_kernel) normally serve only large convolutions; ASR_DEBUG nt_wide /
    nt_wide_force (read once per process) route the GEMM and implicit-convolution tests of this file through them"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ASR_DEBUG="nt_wide=%s,nt_wide_force=1" % mode)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-m", "gpu", "-x", "-q",
                          "-k", "test_gemm_nt or test_implicit_conv or test_conv_as_im2col"], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


@pytest.mark.parametrize("rows,cols,dtype,ld", [(1000, 640, BF16, 640), (32000, 3000, BF16, 3000), (777, 29, BF16, 29), (500, 64, F32, 64),
                                                (3000, 128, BF16, 256), (4097, 2048, BF16, 2048), (257, 8192, BF16, 8192), (100, 8200, BF16, 8200)])
def test_colsum_acc(device, rows, cols, dtype, ld):
    """asr_colsum_acc (bias gradients): out[c] += sum_r x[r][c] on every code path (16-byte rows up to 2048 columns, wide rows,
    scalar fall-back, float32, strided rows), accumulated on top of what is there"""
    from asr import _ops
    g = torch.Generator().manual_seed(rows + cols)
    full = torch.randn(rows, ld, generator=g)
    if dtype == BF16:
        full = _bf(full)
    xd = full.to(device, dtype)[:, :cols]
    out = torch.ones(cols, device=device)
    _ops.colsum_acc(xd, out)
    ref = 1.0 + full[:, :cols].double().sum(dim=0)
    tol = 2e-4 * float(full[:, :cols].abs().sum(dim=0).max())
    assert float((out.cpu().double() - ref).abs().max()) <= tol


def test_tn256_kernel_in_a_forced_process():
    """the 256 x 128 LDS-DMA TN kernel normally serves the convolution weight gradients with >= 256 output channels; ASR_DEBUG tn256=2
    (read once per process) routes the plain TN products of this file through it as well"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ASR_DEBUG="tn256=2")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-m", "gpu", "-x", "-q",
                          "-k", "test_gemm_tn_acc or test_implicit_conv"], env=env, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]


def test_forward_recurrence_without_the_ring_in_a_forced_process():
    """the forward hand-off normally goes through the L2-resident ring; ASR_DEBUG fwd_ring=0 (read once per process) polls the payload in the
    bf16 state sequence instead (sentinel fill + pre-touch): the recurrence tests of this file must pass on that path as well"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ASR_DEBUG="fwd_ring=0")
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_kernels_gpu.py"), "-m", "gpu", "-x", "-q",
                          "-k", "test_gru_full_size_forms_agree or (test_gru_step_kernels and (20-32-64-512 or 40-19-48-128))"],
                         env=env, capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]

