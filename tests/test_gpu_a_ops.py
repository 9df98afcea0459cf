"""GPU: each HIP operator (through the C ABI) against a plain PyTorch fp32
reference of the same op, forward and backward."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _ops():
    from vmrframe_amd import ops
    return ops


def _ints(rows, cols, dt, dev):
    return torch.randint(-3, 4, (rows, cols), device=dev).to(dt)


DT16 = [torch.bfloat16, torch.float16]     # the two 16-bit element types share every kernel (VMR_BF16 / VMR_F16)


def _code(dt):
    from vmrframe_amd import _lib as L
    return {torch.bfloat16: L.BF16, torch.float16: L.F16, torch.float32: L.F32}[dt]


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("shape", [(128, 128, 64), (256, 384, 192), (200, 136, 72), (70, 50, 33), (8, 8, 8),
                                   (1, 1, 5), (130, 4, 1024), (60, 96, 400), (300, 20, 24)])
def test_gemm_exact_integer_all_layouts(dev, dt, shape):
    """Asymmetric small-integer operands: every product/sum is exact in bf16xbf16->f32, so the
    MFMA fragment maps, LDS swizzles and transposed reads must reproduce torch bit for bit."""
    ops = _ops()
    M, N, K = shape
    torch.manual_seed(M * 7 + N * 3 + K)
    for ta in (0, 1):
        for tb in (0, 1):
            A = _ints(K, M, dt, dev) if ta else _ints(M, K, dt, dev)
            B = _ints(K, N, dt, dev) if tb else _ints(N, K, dt, dev)
            ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
            out = ops.mm(A, B, ta, tb, out_f32=True)
            assert torch.equal(out, ref), (dt, shape, ta, tb)


@pytest.mark.parametrize("dt", DT16)
@pytest.mark.parametrize("shape", [(512, 256, 128), (768, 384, 192), (9472, 1024, 128), (1024, 1024, 2368),
                                   (9472, 2048, 192)])
def test_gemm_lds_dma_kernels_exact(dev, shape, dt):
    """The interior LDS-DMA kernels (128x128 / 160x128 tiles, all four layouts, and the 512-thread 320x128 tile the
    cost model picks for the two-round [9472 x 2048] x.W^T product), including the ragged last row tile at
    M = 9472 and fused epilogues with dropout."""
    ops = _ops()
    from vmrframe_amd import _lib as L
    M, N, K = shape
    torch.manual_seed(M + N + K)
    DC = _code(dt)
    for ta in (0, 1):
        for tb in (0, 1):
            A = _ints(K, M, dt, dev) if ta else _ints(M, K, dt, dev)
            B = _ints(K, N, dt, dev) if tb else _ints(N, K, dt, dev)
            ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
            out = ops.mm(A, B, ta, tb, out_f32=True)
            assert torch.equal(out, ref), (shape, ta, tb)
    A, B = _ints(M, K, dt, dev), _ints(N, K, dt, dev)
    res = _ints(M, N, dt, dev)
    rs = torch.rand(M, device=dev)
    o32 = torch.empty(M, N, device=dev)
    ops.gemm(A, B, o32, M, N, K, 0, 0, K, K, N, dtype=DC, residual=res, ldr=N, rowscale=rs,
             flags=L.EPI_DROPOUT | L.EPI_OUT_F32 | L.EPI_RESIDUAL | L.EPI_ROWSCALE, drop=(0.25, 5, None))
    mask = ops.dropout_mask(M * N, 0.25, 5, dev).view(M, N)
    ref = ((A.float() @ B.float().t()) * mask + res.float()) * rs[:, None]
    assert torch.allclose(o32, ref, atol=1e-3, rtol=1e-5)
    acc = torch.ones(M, N, device=dev)
    ops.gemm(A, B, acc, M, N, K, 0, 0, K, K, N, dtype=DC, flags=L.EPI_ACCUM, splitk=2 if K >= 256 else 1)
    assert torch.equal(acc - 1, A.float() @ B.float().t())
    # the register-direct bf16 epilogue (and, at [9472 x 1024], the 160-row tile with its ragged last
    # row tile): bias + ReLU + dropout + aux + residual, both B layouts
    bias = torch.randn(N, device=dev)
    mask = ops.dropout_mask(M * N, 0.25, 9, dev).view(M, N)
    for tb, Bm in ((0, B), (1, B.t().contiguous())):
        out = torch.empty(M, N, device=dev, dtype=dt)
        aux = torch.empty_like(out)
        ops.gemm(A, Bm, out, M, N, K, 0, tb, K, Bm.stride(0), N, dtype=DC, bias=bias, residual=res, aux=aux, ldr=N,
                 flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT | L.EPI_RESIDUAL | L.EPI_AUX, drop=(0.25, 9, None))
        h = torch.relu(A.float() @ B.float().t() + bias) * mask
        assert torch.equal(aux, h.to(dt)), (shape, tb)
        assert torch.equal(out, (h + res.float()).to(dt)), (shape, tb)
    # the same epilogue with the mask kept as a bit matrix (VMR_EPI_AUX_BITS), and the backward kernel that reads it
    d = L.GemmDesc()
    d.A, d.B, d.C, d.bias, d.residual = A.data_ptr(), B.data_ptr(), out.data_ptr(), bias.data_ptr(), res.data_ptr()
    d.lda, d.ldb, d.ldc, d.ldr, d.M, d.N, d.K, d.dtype = K, K, N, N, M, N, K, DC
    d.flags = L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT | L.EPI_RESIDUAL
    d.Z1 = d.Z2 = d.splitk = 1
    assert L.lib().vmr_gemm_aux_bits_supported(C.byref(d)) == 1
    bits = torch.zeros(M, N // 8, device=dev, dtype=torch.uint8)
    out2 = torch.empty(M, N, device=dev, dtype=dt)
    ops.gemm(A, B, out2, M, N, K, 0, 0, K, K, N, dtype=DC, bias=bias, residual=res, aux=bits, ldr=N,
             flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT | L.EPI_RESIDUAL | L.EPI_AUX | L.EPI_AUX_BITS, drop=(0.25, 9, None))
    assert torch.equal(out2, (h + res.float()).to(dt))
    want = (h.to(dt) != 0).view(M, N // 8, 8).to(torch.int32)
    want = (want << torch.arange(8, device=dev, dtype=torch.int32)).sum(-1).to(torch.uint8)
    assert torch.equal(bits, want)
    dy = _ints(M, N, dt, dev)
    dz1, dz3 = torch.empty_like(dy), torch.empty_like(dy)
    db1, db3 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    st = L.stream_ptr()
    L.check(L.lib().vmr_relu_bwd_bias(1, dy.data_ptr(), aux.data_ptr(), dz1.data_ptr(), db1.data_ptr(), M, N, N, 2.0, DC,
                                      0.0, 0, None, None, 1.0, st), "mode 1")
    L.check(L.lib().vmr_relu_bwd_bias(3, dy.data_ptr(), bits.data_ptr(), dz3.data_ptr(), db3.data_ptr(), M, N, N, 2.0, DC,
                                      0.0, 0, None, None, 1.0, st), "mode 3")
    assert torch.equal(dz1, dz3) and torch.allclose(db1, db3, rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("dt", DT16)
def test_gemm_dma_on_and_off_exact(dev, dt):
    """vmr_gemm with the LDS-DMA kernels on (vmr_debug_set_gemm_dma 1, the default) and off (0 = the register-staged
    kernel): all four operand layouts, K = 128 .. 1088 (2 .. 17 ring steps, odd counts included), the fused epilogue and
    a split-K slab product, bit-exact on small integers.  (Round 2's extra ring variants -- BK = 32 x 3 / 4 stages -- are
    no longer in the library; the switch refuses anything but 0 / 1.)"""
    ops = _ops()
    from vmrframe_amd import _lib as L
    DC = _code(dt)
    lib = L.lib()
    assert lib.vmr_debug_set_gemm_dma(3) != 0 and lib.vmr_debug_set_gemm_dma(2) != 0
    try:
        for mode in (1, 0):
            assert lib.vmr_debug_set_gemm_dma(mode) == 0
            for (M, N, K) in [(256, 128, 128), (384, 256, 192), (1280, 384, 1088), (128, 128, 320)]:
                for ta, tb in ((0, 0), (1, 1), (0, 1), (1, 0)):
                    torch.manual_seed(M + N + K + 2 * ta + tb)
                    A = _ints(K, M, dt, dev) if ta else _ints(M, K, dt, dev)
                    B = _ints(K, N, dt, dev) if tb else _ints(N, K, dt, dev)
                    ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
                    assert torch.equal(ops.mm(A, B, ta, tb, out_f32=True), ref), (mode, M, N, K, ta, tb)
            M, N, K = 640, 256, 448
            A, B = _ints(M, K, dt, dev), _ints(N, K, dt, dev)
            ref = A.float() @ B.float().t()
            bias, res = torch.randn(N, device=dev), _ints(M, N, dt, dev)
            out = torch.empty(M, N, device=dev, dtype=dt)
            aux = torch.empty_like(out)
            ops.gemm(A, B, out, M, N, K, 0, 0, K, K, N, dtype=DC, bias=bias, residual=res, aux=aux, ldr=N,
                     flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT | L.EPI_RESIDUAL | L.EPI_AUX, drop=(0.25, 9, None))
            mask = ops.dropout_mask(M * N, 0.25, 9, dev).view(M, N)
            h = torch.relu(ref + bias) * mask
            assert torch.equal(aux, h.to(dt)) and torch.equal(out, (h + res.float()).to(dt)), mode
            M, N, K, sk = 256, 384, 1152, 3
            A, B = _ints(K, M, dt, dev), _ints(K, N, dt, dev)
            ws = torch.empty(sk, M, N, device=dev)
            ops.gemm(A, B, ws, M, N, K, 1, 1, M, N, N, dtype=DC, flags=L.EPI_SLAB, splitk=sk)
            assert torch.equal(ws.sum(0), A.float().t() @ B.float()), mode
    finally:
        lib.vmr_debug_set_gemm_dma(-1)


@pytest.mark.parametrize("dt", DT16)
def test_gemm_8phase_kernel_exact(dev, dt):
    """The 256 x 256 8-phase kernel (gemm_p8_body), forced wherever the shape allows: K-contiguous operands (plain,
    ragged last row tile, fused epilogue with 16-byte permuted stores) and the transposed-operand split-K slab layout,
    bit-exact on small integers; then the default policy on a shape the rounds model gives to it."""
    ops = _ops()
    from vmrframe_amd import _lib as L
    DC = _code(dt)
    lib = L.lib()
    try:
        lib.vmr_debug_set_gemm_p8(2)
        for (M, N, K) in [(256, 256, 128), (768, 512, 448), (9472, 1024, 1024), (1288, 256, 256), (264, 512, 128)]:
            torch.manual_seed(M + N + K)
            A, B = _ints(M, K, dt, dev), _ints(N, K, dt, dev)
            ref = A.float() @ B.float().t()
            assert torch.equal(ops.mm(A, B, 0, 0, out_f32=True), ref), (M, N, K)
            assert torch.equal(ops.mm(A, B, 0, 0), ref.to(dt)), (M, N, K)
            bias = torch.randn(N, device=dev)
            res = _ints(M, N, dt, dev)
            out = torch.empty(M, N, device=dev, dtype=dt)
            aux = torch.empty_like(out)
            ops.gemm(A, B, out, M, N, K, 0, 0, K, K, N, dtype=DC, bias=bias, residual=res, aux=aux, ldr=N,
                     flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT | L.EPI_RESIDUAL | L.EPI_AUX, drop=(0.25, 9, None))
            mask = ops.dropout_mask(M * N, 0.25, 9, dev).view(M, N)
            h = torch.relu(ref + bias) * mask
            assert torch.equal(aux, h.to(dt)) and torch.equal(out, (h + res.float()).to(dt)), (M, N, K)
        for (M, N, K, sk) in [(256, 256, 256, 1), (1024, 1024, 2368, 4), (1024, 1024, 9472, 6), (512, 256, 576, 3)]:
            torch.manual_seed(K)
            A, B = _ints(K, M, dt, dev), _ints(K, N, dt, dev)
            ref = A.float().t() @ B.float()
            if sk == 1:
                out = ops.mm(A, B, 1, 1, out_f32=True)
            else:
                ws = torch.empty(sk, M, N, device=dev)
                ops.gemm(A, B, ws, M, N, K, 1, 1, M, N, N, dtype=DC, flags=L.EPI_SLAB, splitk=sk)
                out = ws.sum(0)
            assert torch.equal(out, ref), (M, N, K, sk)
    finally:
        lib.vmr_debug_set_gemm_p8(-1)
    M, N, K = 9472, 3072, 320                     # 444 tiles of 256 x 256: the default policy's choice
    A, B = _ints(M, K, dt, dev), _ints(N, K, dt, dev)
    assert torch.equal(ops.mm(A, B, 0, 0, out_f32=True), A.float() @ B.float().t())


@pytest.mark.parametrize("dt", DT16)
@pytest.mark.parametrize("shape", [(1024, 1024, 2112, 8), (256, 128, 1024, 4), (1024, 1024, 8192, 8)])
def test_gemm_splitk_slabs_colsum(dev, shape, dt):
    """Weight-gradient shape: dW = A^T.B with split-K slabs (plain fp32 partials + vmr_splitk_reduce) and the
    bias gradient (column sums of A^T) riding on the product."""
    ops = _ops()
    from vmrframe_amd import _lib as L
    M, N, K, sk = shape
    torch.manual_seed(K)
    A, B = _ints(K, M, dt, dev), _ints(K, N, dt, dev)
    ws = torch.empty(sk, M, N, device=dev)
    cs = torch.ones(M, device=dev)
    ops.gemm(A, B, ws, M, N, K, 1, 1, M, N, N, dtype=_code(dt), flags=L.EPI_SLAB, splitk=sk, a_colsum=cs)
    dst = torch.ones(M, N, device=dev)
    L.check(L.lib().vmr_splitk_reduce(ws.data_ptr(), dst.data_ptr(), sk, M * N, N, N, L.stream_ptr()), "reduce")
    assert torch.equal(dst - 1, A.float().t() @ B.float())
    assert torch.equal(cs - 1, A.float().sum(0))
    # slab rows wider than the destination (a weight whose K was zero-padded for the product: 500 -> 512): ld_dst < cols
    # reduces the first ld_dst columns of every slab row into a dense [M, ld_dst] matrix
    if N >= 128:
        kf = N - 12
        dst2 = torch.ones(M, kf, device=dev)
        L.check(L.lib().vmr_splitk_reduce(ws.data_ptr(), dst2.data_ptr(), sk, M * N, N, kf, L.stream_ptr()), "reduce (padded K)")
        assert torch.equal(dst2 - 1, (A.float().t() @ B.float())[:, :kf])


@pytest.mark.parametrize("dt", DT16)
@pytest.mark.parametrize("shape", [(512, 256, 256, 256, 256, 1024, 4), (9472, 1024, 128, 1024, 1024, 1024, 4),
                                   (8192, 1024, 192, 1024, 1024, 2048, 8), (200, 256, 256, 256, 256, 1024, 4)])
def test_gemm2_reduce_exact(dev, shape, dt):
    """vmr_gemm2_reduce: an x.W^T product (with bias + residual epilogue), a transposed-operand split-K slab product
    (with the bias-gradient column sums) and the slab reduction of an EARLIER product in ONE launch -- all three
    bit-exact on small-integer operands, for both tile heights; the last shape (M not a multiple of 8 x 16) takes the
    library's fallback of separate launches and must give the same results."""
    import ctypes as C
    from vmrframe_amd import _lib as L
    M1, N1, K1, M2, N2, K2, sk = shape
    torch.manual_seed(M1 + K2)
    A1, B1 = _ints(M1, K1, dt, dev), _ints(N1, K1, dt, dev)
    bias = torch.randint(-2, 3, (N1,), device=dev).float()
    res = _ints(M1, N1, dt, dev)
    C1 = torch.empty(M1, N1, device=dev, dtype=dt)
    A2, B2 = _ints(K2, M2, dt, dev), _ints(K2, N2, dt, dev)
    slabs = torch.empty(sk, M2, N2, device=dev)
    cs = torch.zeros(M2, device=dev)
    old = torch.randint(-4, 5, (3, 96, 64), device=dev).float()          # slabs of an earlier product
    dst = torch.ones(96, 128, device=dev)                                  # their target: a [96, 64] block inside [96, 128]
    d1 = L.GemmDesc(); d2 = L.GemmDesc()
    d1.A, d1.B, d1.C, d1.bias, d1.residual = A1.data_ptr(), B1.data_ptr(), C1.data_ptr(), bias.data_ptr(), res.data_ptr()
    d1.lda, d1.ldb, d1.ldc, d1.ldr = K1, K1, N1, N1
    d1.M, d1.N, d1.K, d1.transA, d1.transB, d1.dtype = M1, N1, K1, 0, 0, _code(dt)
    d1.flags, d1.alpha, d1.Z1, d1.Z2, d1.splitk = L.EPI_BIAS | L.EPI_RESIDUAL, 1.0, 1, 1, 1
    d2.A, d2.B, d2.C = A2.data_ptr(), B2.data_ptr(), slabs.data_ptr()
    d2.lda, d2.ldb, d2.ldc = M2, N2, N2
    d2.M, d2.N, d2.K, d2.transA, d2.transB, d2.dtype = M2, N2, K2, 1, 1, _code(dt)
    d2.flags, d2.alpha, d2.Z1, d2.Z2, d2.splitk, d2.a_colsum = L.EPI_SLAB, 1.0, 1, 1, sk, cs.data_ptr()
    L.check(L.lib().vmr_gemm2_reduce(C.byref(d1), C.byref(d2), old.data_ptr(), dst.data_ptr(), 3, 96 * 64, 64, 128,
                                      L.stream_ptr()), "vmr_gemm2_reduce")
    assert torch.equal(C1.float(), (A1.float() @ B1.float().t() + bias + res.float()).to(dt).float())
    assert torch.equal(slabs.sum(0), A2.float().t() @ B2.float())
    assert torch.equal(cs, A2.float().sum(0))
    ref = torch.ones(96, 128, device=dev)
    ref[:, :64] += old.sum(0)
    assert torch.equal(dst, ref)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
def test_gemm_epilogue_splitk_dropout(dev, dt):
    ops = _ops()
    from vmrframe_amd import _lib as L
    M, N, K = 192, 256, 128
    A, B = _ints(M, K, dt, dev), _ints(N, K, dt, dev)
    bias = torch.randn(N, device=dev)
    res = _ints(M, N, dt, dev)
    aux = torch.empty(M, N, device=dev, dtype=dt)
    out = torch.empty(M, N, device=dev, dtype=dt)
    ops.gemm(A, B, out, M, N, K, 0, 0, K, K, N, dtype=L.dtype_code(A), bias=bias, residual=res, aux=aux, ldr=N,
             flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_RESIDUAL | L.EPI_AUX)
    h = torch.relu(A.float() @ B.float().t() + bias)
    tol = 1.0 if dt in DT16 else 1e-4   # bf16 output rounding of values up to ~200
    assert (aux.float() - h).abs().max() <= tol
    assert (out.float() - (h + res.float())).abs().max() <= tol
    acc = torch.ones(M, N, device=dev)
    ops.gemm(A, B, acc, M, N, K, 0, 0, K, K, N, dtype=L.dtype_code(A), flags=L.EPI_ACCUM, splitk=2)
    assert torch.equal(acc - 1, A.float() @ B.float().t())
    o32 = torch.empty(M, N, device=dev)
    ops.gemm(A, B, o32, M, N, K, 0, 0, K, K, N, dtype=L.dtype_code(A), flags=L.EPI_DROPOUT | L.EPI_OUT_F32,
             drop=(0.25, 123, None))
    mask = ops.dropout_mask(M * N, 0.25, 123, dev).view(M, N)
    assert torch.allclose(o32, (A.float() @ B.float().t()) * mask, atol=1e-4)
    assert abs(float((mask > 0).float().mean()) - 0.75) < 0.02


def _close(a, b, tol, what=""):
    err = (a.float() - b.float()).abs().max().item()
    scale = max(1.0, b.float().abs().max().item())
    assert err <= tol * scale, f"{what}: err {err} scale {scale}"


def test_linear_fwd_bwd(dev):
    ops = _ops()
    torch.manual_seed(0)
    M, K, N = 300, 500, 64      # K not a multiple of 8 -> padded path
    cache = ops.WeightCache()
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, 1, device=dev, requires_grad=True)
    b = torch.randn(N, device=dev, requires_grad=True)
    res = torch.randn(M, N, device=dev, requires_grad=True)
    xin = x.clone().requires_grad_(True)
    y = ops.linear(ops.to_dtype(xin, torch.float32, pad8=True), W, b, cache, relu=True, residual=res)
    ref = torch.relu(x @ W[:, :, 0].t() + b) + res
    _close(y, ref, 1e-4, "linear fwd")
    g = torch.randn_like(ref)
    gx, gW, gb, gr = torch.autograd.grad(y, [xin, W, b, res], g)
    x2 = x.clone().requires_grad_(True)
    ref2 = torch.relu(x2 @ W[:, :, 0].t() + b) + res
    rx, rW, rb, rr = torch.autograd.grad(ref2, [x2, W, b, res], g)
    _close(gx, rx, 1e-4, "dx"); _close(gW, rW, 1e-4, "dW"); _close(gb, rb, 1e-4, "db"); _close(gr, rr, 1e-5, "dres")


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2), (torch.float16, 3e-2)])
@pytest.mark.parametrize("rows,D,S", [(96, 256, 12), (9472, 1024, 128), (4101, 2048, 3), (7, 512, 7), (130, 1536, 13)])
def test_layernorm_fwd_bwd(dev, dt, tol, rows, D, S):
    """(D = 2048 / 1536: the four- and three-chunk instantiations of ln_bwd, whose software prefetch keeps the most
    registers live across its hand-written wait; odd row counts: the two-rows-per-wave forward's tail)"""
    ops = _ops()
    torch.manual_seed(1)
    cache = ops.WeightCache()
    x = torch.randn(rows, D, device=dev).to(dt).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(D, device=dev)).requires_grad_(True)
    beta = (0.1 * torch.randn(D, device=dev)).requires_grad_(True)
    pos = torch.randn(max(16, S), D, device=dev, requires_grad=True)
    y = ops.layer_norm(x, gamma, beta, 1e-6, cache, pos=pos, S=S)
    xr = x.detach().float().requires_grad_(True)
    posr = pos.detach().to(dt).float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gamma, beta, 1e-6) + posr[:S].repeat(rows // S, 1)
    _close(y, ref, tol, "ln fwd")
    g = torch.randn(rows, D, device=dev)
    gx, gg, gb, gp = torch.autograd.grad(y, [x, gamma, beta, pos], g.to(dt))
    rx, rg, rb, rp = torch.autograd.grad(ref, [xr, gamma, beta, posr], g.to(dt).float())
    _close(gx, rx, tol, "ln dx"); _close(gg, rg, tol, "dgamma"); _close(gb, rb, tol, "dbeta"); _close(gp, rp, tol, "dpos")


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2), (torch.float16, 3e-2)])
@pytest.mark.parametrize("segs", [[(3, 16), (3, 6)], [(2, 128), (2, 20)], [(1, 1), (2, 3)], [(2, 70)], [(2, 40), (3, 9), (1, 130)]])
def test_ln_dwconv_fwd_bwd(dev, dt, tol, segs):
    ops = _ops()
    torch.manual_seed(2)
    D = 64
    rows = sum(b * s for b, s in segs)
    x = torch.randn(rows, D, device=dev).to(dt).requires_grad_(True)
    gamma = (1 + 0.1 * torch.randn(D, device=dev)).requires_grad_(True)
    beta = (0.1 * torch.randn(D, device=dev)).requires_grad_(True)
    w = (torch.randn(D, 1, 7, device=dev) / math.sqrt(7)).requires_grad_(True)
    u = ops.ln_dwconv(x, gamma, beta, w, 1e-6, segs)
    xr = x.detach().float().requires_grad_(True)
    outs, r = [], 0
    for (B, S) in segs:
        n = torch.nn.functional.layer_norm(xr[r:r + B * S].view(B, S, D), (D,), gamma, beta, 1e-6)
        if dt in DT16:
            n = n + (n.to(dt).float() - n).detach()    # the kernel stages LN(x) in bf16
        c = torch.nn.functional.conv1d(n.transpose(1, 2), w, padding=3, groups=D).transpose(1, 2)
        outs.append(c.reshape(B * S, D)); r += B * S
    ref = torch.cat(outs, 0)
    _close(u, ref, tol, "ln_dwconv fwd")
    g = torch.randn(rows, D, device=dev).to(dt)
    gx, gg, gb, gw = torch.autograd.grad(u, [x, gamma, beta, w], g)
    rx, rg, rb, rw = torch.autograd.grad(ref, [xr, gamma, beta, w], g.float())
    _close(gx, rx, tol, "dx"); _close(gg, rg, tol, "dgamma"); _close(gb, rb, tol, "dbeta"); _close(gw, rw, tol, "dw")


def _ref_dual(qkv, kv, vmask, tmask, B, T, Lq, H):
    D = qkv.shape[1] // 3
    hd = D // H
    Nv = B * T
    so, xo = [], []
    for (r0, Lf, t0, Lt, fm, tm) in ((0, T, Nv, Lq, vmask, tmask), (Nv, Lq, 0, T, tmask, vmask)):
        f = qkv[r0:r0 + B * Lf].view(B, Lf, 3, H, hd)
        t = kv[t0:t0 + B * Lt].view(B, Lt, 2, H, hd)
        q, kf, vf = (f[:, :, i].permute(0, 2, 1, 3) for i in range(3))
        kt, vt = (t[:, :, i].permute(0, 2, 1, 3) for i in range(2))
        sm = (fm[:, :, None] * fm[:, None, :])[:, None]
        xm = (fm[:, :, None] * tm[:, None, :])[:, None]
        s = torch.softmax(q @ kf.transpose(-1, -2) / math.sqrt(hd) + (1 - sm) * -1e30, -1)
        x = torch.softmax(q @ kt.transpose(-1, -2) / math.sqrt(hd) + (1 - xm) * -1e30, -1)
        so.append((s @ vf).permute(0, 2, 1, 3).reshape(B * Lf, D))
        xo.append((x @ vt).permute(0, 2, 1, 3).reshape(B * Lf, D))
    return torch.cat(so), torch.cat(xo)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 4e-2), (torch.float16, 4e-2)])
@pytest.mark.parametrize("dims", [(3, 16, 6, 32, 4), (2, 128, 20, 256, 4), (2, 128, 20, 1024, 4), (3, 70, 20, 512, 4),
                                  (2, 40, 33, 512, 4), (2, 1, 1, 256, 2)])
def test_dual_attention_fwd_bwd(dev, dt, tol, dims):
    ops = _ops()
    B, T, Lq, D, H = dims
    torch.manual_seed(3)
    N = B * (T + Lq)
    qkv = torch.randn(N, 3 * D, device=dev).to(dt).requires_grad_(True)
    kv = torch.randn(N, 2 * D, device=dev).to(dt).requires_grad_(True)
    vlen = torch.randint(1, T + 1, (B,), device=dev); vlen[0] = T
    tlen = torch.randint(1, Lq + 1, (B,), device=dev)
    vmask = (torch.arange(T, device=dev)[None] < vlen[:, None]).float()
    tmask = (torch.arange(Lq, device=dev)[None] < tlen[:, None]).float()
    so, xo = ops.dual_attention(qkv, kv, vmask, tmask, B, T, Lq, H)
    qr, kr = qkv.detach().float().requires_grad_(True), kv.detach().float().requires_grad_(True)
    rso, rxo = _ref_dual(qr, kr, vmask, tmask, B, T, Lq, H)
    _close(so, rso, tol, "self ctx"); _close(xo, rxo, tol, "cross ctx")
    g1, g2 = torch.randn_like(rso), torch.randn_like(rxo)
    gq, gk = torch.autograd.grad([so, xo], [qkv, kv], [g1.to(dt), g2.to(dt)])
    rq, rk = torch.autograd.grad([rso, rxo], [qr, kr], [g1.to(dt).float(), g2.to(dt).float()])
    _close(gq, rq, tol, "dqkv"); _close(gk, rk, tol, "dkv")


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 4e-2), (torch.float16, 4e-2)])
@pytest.mark.parametrize("dims", [(5, 12, 64, 4), (64, 9, 512, 2), (37, 5, 512, 4), (70, 3, 256, 1)])
def test_batch_axis_attention_matches_torch_mha(dev, dt, tol, dims):
    """Against nn.MultiheadAttention itself, fed exactly like the reference
    (models/layers.py:567-574): seq-first on [B,T,D] with a float key_padding_mask."""
    ops = _ops()
    torch.manual_seed(4)
    B, T, D, H = dims
    mha = torch.nn.MultiheadAttention(D, H).to(dev).eval()
    x = torch.randn(B, T, D, device=dev)
    vlen = torch.randint(1, T + 1, (B,), device=dev); vlen[0] = T
    vmask = (torch.arange(T, device=dev)[None] < vlen[:, None]).float()
    with torch.no_grad():
        ref = mha(x, x, x, vmask.T)[0]
    cache = ops.WeightCache()
    xin = x.reshape(B * T, D).to(dt)
    qkv = ops.linear(xin, mha.in_proj_weight, mha.in_proj_bias, cache)
    ctx = ops.batch_axis_attention(qkv, vmask, B, T, H)
    out = ops.linear(ctx, mha.out_proj.weight, mha.out_proj.bias, cache)
    _close(out.view(B, T, D), ref, tol, "batch-axis MHA")
    # backward through the core against autograd of a hand-written reference
    qkv2 = torch.randn(B * T, 3 * D, device=dev).to(dt).requires_grad_(True)
    o = ops.batch_axis_attention(qkv2, vmask, B, T, H)
    qr = qkv2.detach().float().requires_grad_(True)
    hd = D // H
    q, k, v = (qr.view(B, T, 3, H, hd)[:, :, i].permute(1, 2, 0, 3) for i in range(3))
    s = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(hd) + vmask.t()[:, None, None, :], -1)
    oref = (s @ v).permute(2, 0, 1, 3).reshape(B * T, D)
    _close(o, oref, tol, "core fwd")
    g = torch.randn_like(oref)
    (gq,) = torch.autograd.grad(o, qkv2, g.to(dt))
    (rq,) = torch.autograd.grad(oref, qr, g.to(dt).float())
    _close(gq, rq, tol, "core bwd")


def test_attention_dropout_is_consistent(dev):
    """With dropout the backward must regenerate the forward's masks: check the gradient of a
    linear functional against finite differences of the SAME (seeded) stochastic function."""
    ops = _ops()
    torch.manual_seed(5)
    B, T, Lq, D, H = 2, 8, 4, 32, 4
    N = B * (T + Lq)
    qkv = torch.randn(N, 3 * D, device=dev, requires_grad=True)
    kv = torch.randn(N, 2 * D, device=dev, requires_grad=True)
    vmask, tmask = torch.ones(B, T, device=dev), torch.ones(B, Lq, device=dev)
    drops = [(0.3, 11 + i, None) for i in range(4)]
    w1, w2 = torch.randn(N, D, device=dev), torch.randn(N, D, device=dev)

    def f(a, b):
        so, xo = ops.dual_attention(a, b, vmask, tmask, B, T, Lq, H, drops)
        return (so * w1).sum() + (xo * w2).sum()
    y = f(qkv, kv)
    gq, gk = torch.autograd.grad(y, [qkv, kv])
    d1, d2 = torch.randn_like(qkv), torch.randn_like(kv)
    eps = 1e-2
    with torch.no_grad():
        fd = (f(qkv + eps * d1, kv + eps * d2) - f(qkv - eps * d1, kv - eps * d2)) / (2 * eps)
    an = (gq * d1).sum() + (gk * d2).sum()
    assert abs(fd.item() - an.item()) <= 2e-2 * max(1.0, abs(an.item())), (fd.item(), an.item())


@pytest.mark.parametrize("dims", [(2, 128, 20, 1024, 4), (3, 70, 20, 512, 4), (2, 64, 64, 512, 4)])
@pytest.mark.parametrize("p", [0.0, 0.2])
def test_fused_attention_equals_composed_path(dev, dims, p):
    """csrc/attention.hip against the three-launch path (GEMM + vmr_softmax_fwd + GEMM) on the same
    inputs and the SAME dropout stream: identical keep pattern, probabilities and contexts to bf16
    rounding; the backward consumes either's saved P unchanged."""
    ops = _ops()
    B, T, Lq, D, H = dims
    torch.manual_seed(13)
    N = B * (T + Lq)
    qkv = torch.randn(N, 3 * D, device=dev).to(torch.bfloat16).requires_grad_(True)
    kv = torch.randn(N, 2 * D, device=dev).to(torch.bfloat16).requires_grad_(True)
    vlen = torch.randint(1, T + 1, (B,), device=dev); vlen[0] = T
    tlen = torch.randint(1, Lq + 1, (B,), device=dev)
    vmask = (torch.arange(T, device=dev)[None] < vlen[:, None]).float()
    tmask = (torch.arange(Lq, device=dev)[None] < tlen[:, None]).float()
    drops = [(p, 21 + i, None) for i in range(4)]
    g1 = torch.randn(N, D, device=dev).to(torch.bfloat16)
    g2 = torch.randn(N, D, device=dev).to(torch.bfloat16)
    res = {}
    old = ops.FUSED_ATTENTION, ops.FUSED_ATTENTION_BWD
    try:
        for fused in (True, False):      # fused forward AND backward kernels vs the GEMM + softmax launches
            ops.FUSED_ATTENTION = ops.FUSED_ATTENTION_BWD = fused
            so, xo = ops.dual_attention(qkv, kv, vmask, tmask, B, T, Lq, H, drops)
            gq, gk = torch.autograd.grad([so, xo], [qkv, kv], [g1, g2])
            res[fused] = (so, xo, gq, gk)
    finally:
        ops.FUSED_ATTENTION, ops.FUSED_ATTENTION_BWD = old
    assert ops.L.lib().vmr_attention_fwd_supported(D // H, T, 1)
    for a, b, what in zip(res[True], res[False], ("self ctx", "cross ctx", "dqkv", "dkv")):
        _close(a, b, 1.5e-2, "fused vs composed " + what)
    if p > 0:   # same keep pattern: an element dropped by one path is dropped by the other
        za, zb = res[True][0] == 0, res[False][0] == 0
        assert (za == zb).float().mean().item() > 0.999


def test_fused_attention_rejects_unsupported_shapes(dev):
    ops = _ops()
    lib = ops.L.lib()
    assert lib.vmr_attention_fwd_supported(256, 128, 1) == 1
    assert lib.vmr_attention_fwd_supported(256, 129, 1) == 0
    assert lib.vmr_attention_fwd_supported(64, 20, 1) == 0
    assert lib.vmr_attention_fwd_supported(256, 20, 0) == 0   # fp32 goes through the composed path
    assert lib.vmr_attention_bwd_supported(256, 128, 128, 1) == 1
    assert lib.vmr_attention_bwd_supported(256, 129, 20, 1) == 0
    assert lib.vmr_attention_bwd_supported(64, 20, 20, 1) == 0


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 2e-2)])
@pytest.mark.parametrize("dims", [(5, 20, 64), (3, 1, 8), (2, 300, 1024)])
def test_weighted_pool_fwd_bwd(dev, dt, tol, dims):
    """vmr_weighted_pool_fwd/bwd against the reference formulation (models/layers.py:440-453)."""
    ops = _ops()
    B, Ls, D = dims
    torch.manual_seed(17)
    x = torch.randn(B, Ls, D, device=dev).to(dt).requires_grad_(True)
    w = (torch.randn(D, 1, device=dev) / math.sqrt(D)).requires_grad_(True)
    lens = torch.randint(1, Ls + 1, (B,), device=dev); lens[0] = Ls
    mask = (torch.arange(Ls, device=dev)[None] < lens[:, None]).float()
    out = ops.weighted_pool(x, w, mask)
    xr = x.detach().float().requires_grad_(True)
    wr = w.detach().clone().requires_grad_(True)
    alpha = torch.softmax(torch.tensordot(xr, wr, dims=1) + (1.0 - mask.unsqueeze(2)) * -1e30, dim=1)
    ref = torch.matmul(xr.transpose(1, 2), alpha).squeeze(2)
    _close(out, ref, tol, "pooled")
    g = torch.randn_like(ref)
    gx, gw = torch.autograd.grad(out, [x, w], g.to(dt))
    rx, rw = torch.autograd.grad(ref, [xr, wr], g.to(dt).float())
    _close(gx, rx, tol, "dx"); _close(gw, rw, tol, "dw")


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 3e-2), (torch.float16, 3e-2)])
def test_linear_column_slices_and_row_broadcast_residual(dev, dt, tol):
    """Conv1D(cat([a, pooled.expand], -1)) (reference CQConcatenate, layers.py:462-468) as two GEMMs on
    column slices of the one weight, the pooled half broadcast over each clip's rows by the epilogue."""
    ops = _ops()
    torch.manual_seed(19)
    B, T, D, N = 3, 128, 256, 128
    cache = ops.WeightCache()
    a = torch.randn(B * T, D, device=dev).to(dt).requires_grad_(True)
    pooled = torch.randn(B, D, device=dev).to(dt).requires_grad_(True)
    W = (torch.randn(N, 2 * D, 1, device=dev) / math.sqrt(2 * D)).requires_grad_(True)
    bias = torch.randn(N, device=dev, requires_grad=True)
    pq = ops.linear(pooled, W, bias, cache, kslice=(D, 2 * D))
    y = ops.linear(a, W, None, cache, kslice=(0, D), residual=pq, res_div=T)
    ar, pr = a.detach().float().requires_grad_(True), pooled.detach().float().requires_grad_(True)
    Wr, br = W.detach().clone().requires_grad_(True), bias.detach().clone().requires_grad_(True)
    cat = torch.cat([ar.view(B, T, D), pr[:, None, :].expand(B, T, D)], 2).reshape(B * T, 2 * D)
    Wm = Wr.view(N, 2 * D)
    if dt in DT16:
        Wm = Wm + (Wm.to(dt).float() - Wm).detach()
    ref = cat @ Wm.t() + br
    _close(y, ref, tol, "y")
    g = torch.randn_like(ref)
    ga, gp, gW, gb = torch.autograd.grad(y, [a, pooled, W, bias], g.to(dt))
    ra, rp, rW, rb = torch.autograd.grad(ref, [ar, pr, Wr, br], g.to(dt).float())
    _close(ga, ra, tol, "da"); _close(gp, rp, tol, "dpooled"); _close(gW, rW, tol, "dW"); _close(gb, rb, tol, "db")


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-4), (torch.bfloat16, 2e-2), (torch.float16, 2e-2)])
@pytest.mark.parametrize("dims", [(8192, 1, 1024), (8192, 4, 1024), (50, 8, 32), (3, 3, 2048), (129, 1, 8)])
def test_narrow_linear_fwd_bwd(dev, dt, tol, dims):
    """The N <= 8 output heads (match N=4, start/end N=1) on the matrix-vector kernels vs torch."""
    ops = _ops()
    M, N, K = dims
    torch.manual_seed(23)
    x = torch.randn(M, K, device=dev).to(dt).requires_grad_(True)
    W = (torch.randn(N, K, 1, device=dev) / math.sqrt(K)).requires_grad_(True)
    b = torch.randn(N, device=dev, requires_grad=True)
    y = ops.narrow_linear(x, W, b)
    assert y.dtype == torch.float32 and y.shape == (M, N)
    xr = x.detach().float().requires_grad_(True)
    Wr, br = W.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = xr @ Wr[:, :, 0].t() + br
    _close(y, ref, 1e-4 if dt == torch.float32 else 1e-3, "y")     # fp32 accumulation of exact bf16 inputs
    g = torch.randn_like(ref)
    gx, gW, gb = torch.autograd.grad(y, [x, W, b], g)
    rx, rW, rb = torch.autograd.grad(ref, [xr, Wr, br], g)
    _close(gx, rx, tol, "dx"); _close(gW, rW, 2e-4 if dt == torch.float32 else 2e-3, "dW"); _close(gb, rb, 1e-4, "db")


def test_gumbel_softmax_and_match_loss(dev):
    """vmr_gumbel_softmax_* and vmr_match_loss_* against torch (explicit noise), and the statistics of the
    in-kernel noise (reference models/SeqPAN.py:79, models/loss.py:24-41)."""
    ops = _ops()
    torch.manual_seed(29)
    R, Cc, D = 777, 4, 64
    logits = torch.randn(R, Cc, device=dev, requires_grad=True)
    noise = -torch.empty(R, Cc, device=dev).exponential_().log()
    E = torch.randn(D, Cc, device=dev, requires_grad=True)
    labels = torch.randint(0, Cc, (R,), device=dev)
    vmask = (torch.rand(R, device=dev) > 0.3).float()
    probs, padded = ops.gumbel_softmax(logits, noise, 0.3, 1, None, 8, torch.bfloat16)
    lr = logits.detach().clone().requires_grad_(True)
    Er = E.detach().clone().requires_grad_(True)
    pref = torch.softmax((lr + noise) / 0.3, dim=-1)
    _close(probs, pref, 1e-5, "probs")
    assert padded.shape == (R, 8) and (padded[:, 4:] == 0).all()
    _close(padded[:, :4], pref, 1e-2, "padded")
    loss = ops.match_loss(probs.view(1, R, Cc), E, labels.view(1, R), vmask.view(1, R))
    onehot = torch.nn.functional.one_hot(labels, Cc).float()
    lref = (-(onehot * pref).sum(-1) * vmask).sum() / (vmask.sum() + 1e-12)
    lref = lref + torch.norm(Er.t() @ Er * (1.0 - torch.eye(Cc, device=dev)), p=2)
    assert abs(loss.item() - lref.item()) <= 1e-4 * max(1.0, abs(lref.item()))
    w = torch.randn(R, 8, device=dev)
    total = loss * 1.7 + (padded.float() * w).sum()
    tref = lref * 1.7 + (pref * w[:, :4]).sum()
    g1 = torch.autograd.grad(total, [logits, E])
    g2 = torch.autograd.grad(tref, [lr, Er])
    _close(g1[0], g2[0], 2e-2, "dlogits")      # (the padded branch carries bf16-rounded probabilities)
    _close(g1[1], g2[1], 1e-4, "dlabel_embs")
    # in-kernel noise: Gumbel(0,1) has mean 0.5772 and variance pi^2/6; draws differ per seed / step
    z = torch.zeros(200000, 1, device=dev)
    tau = 1.0
    # the in-kernel uniform must stay strictly inside (0, 1): over 2^26 draws a 24-bit construction hits U = 1.0
    # (g = +inf, NaN probabilities) about four times -- a 200-step BaseFast run did at its replay 32
    big, _ = ops.gumbel_softmax(torch.zeros(1 << 24, 4, device=dev), None, tau, 12345, None, 8, torch.bfloat16)
    assert bool(torch.isfinite(big).all())
    del big
    p1, _ = ops.gumbel_softmax(torch.zeros(50000, 4, device=dev), None, tau, 7, None, 8, torch.bfloat16)
    p2, _ = ops.gumbel_softmax(torch.zeros(50000, 4, device=dev), None, tau, 8, None, 8, torch.bfloat16)
    assert not torch.equal(p1, p2)
    assert abs(p1.mean().item() - 0.25) < 1e-6 + 1e-3           # rows sum to one
    # with equal logits every class wins equally often
    wins = torch.bincount(p1.argmax(1), minlength=4).float() / p1.shape[0]
    assert (wins - 0.25).abs().max().item() < 0.01


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2), (torch.float16, 2e-2)])
def test_scale_shift_fwd_bwd(dev, dt, tol):
    ops = _ops()
    torch.manual_seed(31)
    R, D = 1280, 1024
    x = torch.randn(2, R // 2, D, device=dev).to(dt).requires_grad_(True)
    a = torch.randn(1, 1, D, device=dev, requires_grad=True)
    b = torch.randn(D, 1, device=dev, requires_grad=True)
    y = ops.scale_shift(x, a, b)
    xr = x.detach().float().requires_grad_(True)
    ar, br = a.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    ref = xr * ar + br.view(1, 1, D)
    _close(y, ref, tol, "y")
    g = torch.randn_like(ref)
    gx, ga, gb = torch.autograd.grad(y, [x, a, b], g.to(dt))
    rx, ra, rb = torch.autograd.grad(ref, [xr, ar, br], g.to(dt).float())
    _close(gx, rx, tol, "dx"); _close(ga, ra, tol, "da"); _close(gb, rb, tol, "db")


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2), (torch.float16, 3e-2)])
@pytest.mark.parametrize("dims", [(7, 5, 16, 12), (64 * 20, 8, 100, 60), (33, 16, 24, 9)])
def test_char_cnn_fwd_bwd(dev, dt, tol, dims):
    """csrc/charcnn.hip against the reference formulation of CharacterEmbedding (models/layers.py:51-75):
    nn.Embedding(padding_idx=0) -> Conv2d(char_dim, 10k, (1,k)) + ReLU -> max over positions, k = 1..4."""
    ops = _ops()
    Wn, Cc, CD, NCH = dims
    torch.manual_seed(37)
    ids = torch.randint(0, NCH, (Wn, Cc), device=dev)
    ids[0] = 0                                             # an all-padding word
    table = torch.randn(NCH, CD, device=dev)
    table[0] = 0
    table.requires_grad_(True)
    ws = [(torch.randn(10 * (k + 1), CD, 1, k + 1, device=dev) / math.sqrt(CD * (k + 1))).requires_grad_(True) for k in range(4)]
    bs = [(0.1 * torch.randn(10 * (k + 1), device=dev)).requires_grad_(True) for k in range(4)]
    out = ops.char_cnn(ids.view(1, Wn, Cc), table, ws, bs, ops.NO_DROP, dt)
    tr = table.detach().clone().requires_grad_(True)
    wr = [w.detach().clone().requires_grad_(True) for w in ws]
    br = [b.detach().clone().requires_grad_(True) for b in bs]
    ce = torch.nn.functional.embedding(ids, tr, padding_idx=0)                      # [W, C, CD]
    if dt in DT16:                                                         # operands rounded like the kernel's
        ce = ce + (ce.to(dt).float() - ce).detach()
    x = ce.permute(1 - 1, 2, 1).unsqueeze(2)                                        # [W, CD, 1, C]
    feats = []
    for k in range(4):
        w = wr[k] if dt == torch.float32 else wr[k] + (wr[k].to(dt).float() - wr[k]).detach()
        y = torch.relu(torch.nn.functional.conv2d(x, w, br[k]))                     # [W, 10k, 1, C-k]
        feats.append(y.max(dim=3).values.squeeze(2))
    ref = torch.cat(feats, 1)
    assert out.shape == ref.shape
    _close(out, ref, tol, "char features")
    g = torch.randn_like(ref)
    g1 = torch.autograd.grad(out, [table, *ws, *bs], g.to(dt))
    g2 = torch.autograd.grad(ref, [tr, *wr, *br], g.to(dt).float())
    for a, b, name in zip(g1, g2, ["dtable"] + [f"dW{k}" for k in range(4)] + [f"db{k}" for k in range(4)]):
        _close(a, b, tol, name)
    assert (g1[0][0] == 0).all()                                                     # padding row: no gradient


def test_char_cnn_dropout_is_consistent(dev):
    """Dropout on the gathered character rows: the backward regenerates the forward's mask (finite differences
    of the same seeded function)."""
    ops = _ops()
    torch.manual_seed(41)
    Wn, Cc, CD, NCH = 40, 8, 20, 15
    ids = torch.randint(1, NCH, (Wn, Cc), device=dev)
    table = torch.randn(NCH, CD, device=dev, requires_grad=True)
    ws = [(torch.randn(10 * (k + 1), CD, 1, k + 1, device=dev) / math.sqrt(CD * (k + 1))).requires_grad_(True) for k in range(4)]
    bs = [(0.5 + 0.1 * torch.randn(10 * (k + 1), device=dev)).requires_grad_(True) for k in range(4)]   # keep ReLUs active
    w8 = torch.randn(Wn, 100, device=dev)
    drop = (0.3, 77, None)

    def f(t):
        return (ops.char_cnn(ids, t, ws, bs, drop, torch.float32) * w8).sum()
    y = f(table)
    (gt,) = torch.autograd.grad(y, [table])
    d = torch.randn_like(table)
    eps = 1e-3
    with torch.no_grad():
        fd = (f(table + eps * d) - f(table - eps * d)) / (2 * eps)
    an = (gt * d).sum()
    assert abs(fd.item() - an.item()) <= 3e-2 * max(1.0, abs(an.item())), (fd.item(), an.item())
    a, b = ops.char_cnn(ids, table, ws, bs, drop, torch.float32), ops.char_cnn(ids, table, ws, bs, (0.3, 78, None), torch.float32)
    assert not torch.equal(a, b)


@pytest.mark.parametrize("shape", [(300, 500, 512), (64, 1024, 1024), (37, 20, 24)])
def test_cast_pad_dropout_matches_scalar_definition(dev, shape):
    """vmr_cast (vector path: 8 columns per thread) = zero-padded cast with the counter-based input dropout
    (reference models/layers.py:120), checked against the mask generator of the same stream."""
    ops = _ops()
    rows, cols, ld = shape
    torch.manual_seed(43)
    x = torch.randn(rows, cols, device=dev)
    out = ops.cast_pad(x, torch.bfloat16, (0.25, 99, None), mult=64 if ld % 64 == 0 else 8)
    assert out.shape == (rows, ld)
    mask = ops.dropout_mask(rows * cols, 0.25, 99, dev).view(rows, cols)
    ref = (x * mask).to(torch.bfloat16)
    assert torch.equal(out[:, :cols], ref)
    assert (out[:, cols:] == 0).all()
    plain = ops.cast_pad(x, torch.bfloat16, ops.NO_DROP, mult=64 if ld % 64 == 0 else 8)
    assert torch.equal(plain[:, :cols], x.to(torch.bfloat16))


def _ceil8(v):
    return (v + 7) // 8 * 8


@pytest.mark.parametrize("dims", [(64, 128, 20, 1024), (3, 70, 9, 256), (2, 128, 32, 512), (2, 1, 1, 256)])
@pytest.mark.parametrize("orient", [0, 1])
@pytest.mark.parametrize("dt", DT16)
def test_cq_score_kernel_matches_composed_path(dev, dims, orient, dt):
    """csrc/cqscore.hip (score + both softmaxes, one launch) against the three-launch path (batched GEMM +
    vmr_cq_softmax_fwd) and against torch, forward and backward, both orientations of CQAttention."""
    ops = _ops()
    B, Ll, Ls, D = dims
    torch.manual_seed(47)
    lng = (torch.randn(B, Ll, D, device=dev) / math.sqrt(D) * 4).to(dt).requires_grad_(True)
    sht = torch.randn(B, Ls, D, device=dev).to(dt).requires_grad_(True)
    term = torch.randn(B, Ls, device=dev, requires_grad=True)
    ll = torch.randint(1, Ll + 1, (B,), device=dev); ll[0] = Ll
    ls = torch.randint(1, Ls + 1, (B,), device=dev); ls[0] = Ls
    ml = (torch.arange(Ll, device=dev)[None] < ll[:, None]).float()
    ms = (torch.arange(Ls, device=dev)[None] < ls[:, None]).float()
    assert ops.cq_score_supported(Ll, Ls, D, dt)
    Sr, Sc = ops.cq_score(lng, sht, term, ml, ms, orient)
    # torch reference in the (context, query) layout of the reference module
    lr, sr, tr = lng.detach().float().requires_grad_(True), sht.detach().float().requires_grad_(True), term.detach().clone().requires_grad_(True)
    M = lr @ sr.transpose(1, 2) + tr[:, None, :]                    # [B, v, t]
    Pt = torch.softmax(M + (1 - ms[:, None, :]) * -1e30, dim=2)
    Pv = torch.softmax(M + (1 - ml[:, :, None]) * -1e30, dim=1)
    rr, rc = (Pt, Pv) if orient == 0 else (Pv.transpose(1, 2), Pt.transpose(1, 2))
    _close(Sr, rr, 2e-2, "S_row"); _close(Sc, rc, 2e-2, "S_col")
    g1, g2 = torch.randn_like(rr), torch.randn_like(rc)
    ga = torch.autograd.grad([Sr, Sc], [lng, sht, term], [g1.to(dt), g2.to(dt)])
    gb = torch.autograd.grad([rr, rc], [lr, sr, tr], [g1.to(dt).float(), g2.to(dt).float()])
    for a, b, name in zip(ga, gb, ("dlong", "dshort", "dterm")):
        _close(a, b, 4e-2, name)
    # and the composed HIP path on the same inputs
    if orient == 0:
        S2 = ops.bmm(lng, sht, 0, 0, out_f32=True)
        Cr, Cc = ops.cq_softmax(S2, None, term, ml, ms, dt)
    else:
        S2 = ops.bmm(sht, lng, 0, 0, out_f32=True)
        Cr, Cc = ops.cq_softmax(S2, term, None, ms, ml, dt)
    _close(Sr, Cr, 1e-2, "fused vs composed S_row"); _close(Sc, Cc, 1e-2, "fused vs composed S_col")


@pytest.mark.parametrize("dims", [(64, 128, 20, 1024), (3, 70, 9, 256), (2, 128, 32, 512), (2, 33, 1, 256), (5, 128, 17, 768),
                                  (2, 16, 16, 256), (3, 2, 1, 256), (2, 100, 24, 1024),
                                  (2, 256, 20, 512), (3, 200, 9, 256), (64, 256, 20, 1024), (2, 129, 32, 256)])
@pytest.mark.parametrize("orient", [0, 1])
@pytest.mark.parametrize("dt", DT16)
def test_fused_cq_block_matches_torch(dev, dims, orient, dt):
    """csrc/cqapply.hip + cqscore.hip: the whole CQAttention core (trilinear score, both masked softmaxes, c2q,
    q2c = S_.(S_t^T.C), the 4-way concat; reference models/layers.py:417-424) forward and backward against fp32 torch
    autograd on the same bf16 inputs, both directions (context = video / context = query), ragged lengths, and the
    composed HIP path (batched GEMMs + cat4) on the same inputs."""
    ops = _ops()
    B, Ll, Ls, D = dims
    torch.manual_seed(53)
    lng = (torch.randn(B, Ll, D, device=dev) / math.sqrt(D) * 4).to(dt).requires_grad_(True)     # the score operands
    sht = torch.randn(B, Ls, D, device=dev).to(dt).requires_grad_(True)
    term = torch.randn(B, Ls, device=dev, requires_grad=True)
    Lc, Lq = (Ll, Ls) if orient == 0 else (Ls, Ll)
    ctx = torch.randn(B, Lc, D, device=dev).to(dt).requires_grad_(True)                            # the apply-stage streams
    qry = torch.randn(B, Lq, D, device=dev).to(dt).requires_grad_(True)
    ll = torch.randint(1, Ll + 1, (B,), device=dev); ll[0] = Ll
    ls = torch.randint(1, Ls + 1, (B,), device=dev); ls[0] = Ls
    ml = (torch.arange(Ll, device=dev)[None] < ll[:, None]).float()
    ms = (torch.arange(Ls, device=dev)[None] < ls[:, None]).float()
    if Lc == Lq and orient == 1:      # equal lengths are orient 0 by definition (cq_block, SeqPAN.cq_attention_core)
        return
    assert ops.cq_block_supported(Lc, Lq, D, dt)
    out = ops.cq_block(ctx, qry, lng, sht, term, ml, ms, orient)
    assert out.shape == (B * Lc, 4 * D)
    f = lambda t: t.detach().float().requires_grad_(True)
    cr, qr, lr, sr, tr = f(ctx), f(qry), f(lng), f(sht), term.detach().clone().requires_grad_(True)
    if orient == 0:
        S = lr @ sr.transpose(1, 2) + tr[:, None, :]
        cmask, qmask = ml, ms
    else:
        S = sr @ lr.transpose(1, 2) + tr[:, :, None]
        cmask, qmask = ms, ml
    S_ = torch.softmax(S + (1 - qmask[:, None, :]) * -1e30, dim=2)
    S_t = torch.softmax(S + (1 - cmask[:, :, None]) * -1e30, dim=1)
    c2q = S_ @ qr
    q2c = (S_ @ S_t.transpose(1, 2)) @ cr                                  # the reference's association (layers.py:423)
    ref = torch.cat([cr, c2q, cr * c2q, cr * q2c], dim=2).reshape(B * Lc, 4 * D)
    _close(out, ref, 2e-2, "cat4")
    g = torch.randn_like(ref).to(dt)
    ga = torch.autograd.grad(out, [ctx, qry, lng, sht, term], g)
    gb = torch.autograd.grad(ref, [cr, qr, lr, sr, tr], g.float())
    for a, b, name in zip(ga, gb, ("dctx", "dqry", "dlong", "dshort", "dterm")):
        _close(a, b, 4e-2, name)
    # the composed path (score kernel + three batched GEMMs + cat4) on the same inputs
    if Ll > 128:        # (its bf16-pair score kernel stops at 128 long rows; the fp32 torch comparison above stands)
        return
    Sp, Stp = ops.cq_score(lng, sht, term, ml, ms, orient)
    c2 = ops.bmm(Sp, qry, 0, 1); mid = ops.bmm(Stp, ctx, 1, 1); q2 = ops.bmm(Sp, mid, 0, 1)
    comp = ops.cat4(ctx.reshape(B * Lc, D), c2.reshape(B * Lc, D), q2.reshape(B * Lc, D))
    _close(out, comp, 1e-2, "fused vs composed cat4")
    gc = torch.autograd.grad(comp, [ctx, qry, lng, sht, term], g)
    for a, b, name in zip(ga, gc, ("dctx", "dqry", "dlong", "dshort", "dterm")):
        _close(a, b, 6e-2, name + " fused vs composed")   # (composed: bf16 probabilities; fused: fp32)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_text_embed_matches_torch(dev, dt):
    """ops.text_embed (vmr_word_embedding_fwd/bwd + vmr_char_cnn_fwd/bwd into one [words, ldo] matrix) against the
    reference's WordEmbedding + CharacterEmbedding composed in torch (models/layers.py:28-75), dropout off: the word
    columns are an exact gather of pad / unk / glove rows, the zero padding is zero, and only unk_vec / the char
    parameters receive gradients."""
    ops = _ops()
    torch.manual_seed(3)
    B, Lq, Cc, nw, nc, wd, CD = 5, 9, 7, 40, 30, 300, 100
    wid = torch.randint(0, nw, (B, Lq), device=dev); wid[0, :3] = torch.tensor([0, 1, 1], device=dev)
    cid = torch.randint(0, nc, (B, Lq, Cc), device=dev)
    pad = torch.zeros(1, wd, device=dev); pad[0, 5] = 0.25                  # (a loaded checkpoint may hold anything)
    unk = torch.randn(1, wd, device=dev, requires_grad=True)
    glove = torch.randn(nw - 2, wd, device=dev)
    table = torch.randn(nc, CD, device=dev, requires_grad=True)
    ws = [torch.randn(10 * (k + 1), CD, 1, k + 1, device=dev, requires_grad=True) for k in range(4)]
    bs = [torch.randn(10 * (k + 1), device=dev, requires_grad=True) for k in range(4)]
    ldo = 400 if dt == torch.float32 else 512
    out = ops.text_embed(wid, cid, pad, unk, glove, table, ws, bs, ops.NO_DROP, ops.NO_DROP, dt, ldo)
    assert out.shape == (B * Lq, ldo) and out.dtype == dt
    ref_w = torch.cat([pad, unk, glove], 0)[wid.reshape(-1)]
    assert torch.equal(out[:, :wd].float(), ref_w.to(dt).float())
    assert float(out[:, 400:].abs().max() if ldo > 400 else 0.0) == 0.0
    ce = torch.nn.functional.embedding(cid, table, padding_idx=0).permute(0, 3, 1, 2)          # [B, CD, L, C]
    feats = [torch.relu(torch.nn.functional.conv2d(ce, w, b)).amax(dim=3).permute(0, 2, 1) for w, b in zip(ws, bs)]
    ref_c = torch.cat(feats, 2).reshape(B * Lq, 100)
    _close(out[:, wd:400], ref_c, 2e-2 if dt in DT16 else 1e-4, "char features")
    g = torch.randn(B * Lq, ldo, device=dev).to(dt)
    ga = torch.autograd.grad(out, [unk, table, *ws, *bs], g)
    ref = torch.cat([ref_w, ref_c], 1)
    gb = torch.autograd.grad(ref, [unk, table, *ws, *bs], g[:, :400].float())
    for a, b, n in zip(ga, gb, ["unk", "table"] + [f"w{k}" for k in range(4)] + [f"b{k}" for k in range(4)]):
        if dt in DT16 and n != "unk":
            continue      # (the bf16 CNN stages bf16-rounded rows / weights: covered, with a rounding-aware reference, by test_char_cnn_fwd_bwd)
        _close(a, b, 1e-2 if dt in DT16 else 1e-3, n)
    # dropout on: the counter mask is regenerated identically in the backward (same keep pattern as the forward)
    d = (0.3, 4242, None)
    o2 = ops.text_embed(wid, cid, pad, unk, glove, table, ws, bs, d, ops.NO_DROP, torch.float32, 400)
    kept = (o2[:, :wd] != 0) | (ref_w == 0)
    assert 0.6 < float(kept.float().mean()) < 0.8
    gu = torch.autograd.grad(o2, unk, torch.ones_like(o2))[0]
    want = (kept[wid.reshape(-1) == 1].float() / 0.7).sum(0, keepdim=True)
    _close(gu, want, 1e-4, "unk gradient under dropout")


def test_soft_ce_matches_torch(dev):
    ops = _ops()
    torch.manual_seed(6)
    B, T = 7, 128
    zs = torch.randn(B, T, device=dev, requires_grad=True)
    ze = torch.randn(B, T, device=dev, requires_grad=True)
    ys, ye = torch.rand(B, T, device=dev) * 3, torch.rand(B, T, device=dev)
    loss = ops.soft_ce(zs, ze, ys, ye)
    ce = torch.nn.CrossEntropyLoss(reduction="mean")
    ref = ce(zs, ys) + ce(ze, ye)
    assert abs(loss.item() - ref.item()) < 1e-4 * max(1, abs(ref.item()))
    g1 = torch.autograd.grad(loss, [zs, ze])
    g2 = torch.autograd.grad(ref, [zs, ze])
    for a, b in zip(g1, g2):
        _close(a, b, 1e-5, "soft_ce grad")


def test_product_path_rejects_cpu_tensors(dev):
    ops = _ops()
    with pytest.raises(RuntimeError):
        ops.soft_ce(torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2, 4), torch.zeros(2, 4))


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 2e-2), (torch.float16, 2e-2)])
@pytest.mark.parametrize("dims", [(3, 16, 6), (2, 128, 20), (2, 20, 128), (1, 1, 1), (2, 7, 300), (2, 300, 3)])
def test_cq_softmax_fwd_bwd(dev, dt, tol, dims):
    """The two masked softmaxes of CQAttention (models/layers.py:419-421) incl. fully masked rows/columns."""
    ops = _ops()
    B, Lc, Lq = dims
    torch.manual_seed(7)
    ld = (Lq + 7) // 8 * 8
    S2buf = torch.randn(B, Lc, ld, device=dev)
    S2 = S2buf[..., :Lq].requires_grad_(True)
    rowt = torch.randn(B, Lc, device=dev, requires_grad=True)
    colt = torch.randn(B, Lq, device=dev, requires_grad=True)
    cmask = (torch.rand(B, Lc, device=dev) > 0.3).float(); cmask[:, 0] = 1
    qmask = (torch.rand(B, Lq, device=dev) > 0.3).float(); qmask[:, 0] = 1
    if B > 1:
        qmask[1] = 0          # a sample whose queries are all padding: uniform softmax, not NaN
    Sr, Sc = ops.cq_softmax(S2, rowt, colt, cmask, qmask, dt)
    S = S2 + rowt[:, :, None] + colt[:, None, :]
    Rr = torch.softmax(S + (1 - qmask[:, None, :]) * -1e30, dim=2)
    Rc = torch.softmax(S + (1 - cmask[:, :, None]) * -1e30, dim=1)
    _close(Sr, Rr, tol, "S_row"); _close(Sc, Rc, tol, "S_col")
    assert torch.isfinite(Sr.float()).all() and torch.isfinite(Sc.float()).all()
    g1, g2 = torch.randn_like(Rr), torch.randn_like(Rc)
    mine = torch.autograd.grad([Sr, Sc], [S2, rowt, colt], [g1.to(dt), g2.to(dt)])
    ref = torch.autograd.grad([Rr, Rc], [S2, rowt, colt], [g1.to(dt).float(), g2.to(dt).float()])
    for a, b, n in zip(mine, ref, ("dS2", "drow", "dcol")):
        _close(a, b, 5 * tol, n)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 2e-2), (torch.float16, 2e-2)])
def test_fused_elementwise_programs(dev, dt, tol):
    """cross gate, sigmoid gate, 4-way CQ concat (models/layers.py:374,380,424) vs torch, fwd + bwd."""
    ops = _ops()
    torch.manual_seed(8)
    N, D = 70, 64
    t = lambda *s: torch.randn(*s, device=dev).to(dt).requires_grad_(True)
    ss, sv, xs, xv = t(N, D), t(N, D), t(N, D), t(N, D)
    out = ops.cross_gate(ss, sv, xs, xv)
    refs = [x.detach().float().requires_grad_(True) for x in (ss, sv, xs, xv)]
    ref = refs[0] * refs[3] + refs[2] * refs[1]
    _close(out, ref, tol, "gate")
    g = torch.randn(N, D, device=dev).to(dt)
    for a, b in zip(torch.autograd.grad(out, [ss, sv, xs, xv], g), torch.autograd.grad(ref, refs, g.float())):
        _close(a, b, tol, "gate grad")
    sv2 = t(N, 2 * D)
    rm = (torch.rand(N, device=dev) > 0.3).float()
    out = ops.sigmoid_gate(sv2, rm)
    r2 = sv2.detach().float().requires_grad_(True)
    ref = torch.sigmoid(r2[:, :D] + -1e30 * (1 - rm[:, None])) * r2[:, D:]
    _close(out, ref, tol, "siggate")
    (a,) = torch.autograd.grad(out, sv2, g)
    (b,) = torch.autograd.grad(ref, r2, g.float())
    _close(a, b, tol, "siggate grad")
    C_, c2q, q2c = t(N, D), t(N, D), t(N, D)
    out = ops.cat4(C_, c2q, q2c)
    rs = [x.detach().float().requires_grad_(True) for x in (C_, c2q, q2c)]
    ref = torch.cat([rs[0], rs[1], rs[0] * rs[1], rs[0] * rs[2]], 1)
    _close(out, ref, tol, "cat4")
    g4 = torch.randn(N, 4 * D, device=dev).to(dt)
    for a, b in zip(torch.autograd.grad(out, [C_, c2q, q2c], g4), torch.autograd.grad(ref, rs, g4.float())):
        _close(a, b, 2 * tol, "cat4 grad")


def test_embedding_gather_scatter(dev):
    ops = _ops()
    torch.manual_seed(9)
    table = torch.randn(30, 100, device=dev, requires_grad=True)
    idx = torch.randint(0, 30, (4, 6, 5), device=dev)
    out = ops.embedding(idx, table, 0)
    ref = torch.nn.functional.embedding(idx, table, padding_idx=0)
    assert torch.equal(out, ref)
    g = torch.randn_like(ref)
    (a,) = torch.autograd.grad(out, table, g)
    (b,) = torch.autograd.grad(ref, table, g)
    _close(a, b, 1e-5, "embedding grad")
    assert float(a[0].abs().max()) == 0.0       # padding_idx row gets no gradient


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_add_pos_matches_torch(dev, dt):
    """ops.add_pos (FeatureEncoderPredict's positional add, reference layers.py:626-631): forward against torch on the
    fp32 table, backward = the incoming gradient for x and its per-position batch sum for the table."""
    ops = _ops()
    B, S, D = 5, 12, 64
    torch.manual_seed(3)
    x = torch.randn(B * S, D, device=dev).to(dt).requires_grad_(True)
    pos = torch.randn(16, D, device=dev, requires_grad=True)
    y = ops.add_pos(x, pos, S)
    ref = (x.detach().float().view(B, S, D) + pos.detach()[:S].view(1, S, D)).reshape(B * S, D)
    assert torch.equal(y, ref.to(dt))
    g = torch.randn(B * S, D, device=dev).to(dt)
    y.backward(g)
    assert torch.equal(x.grad, g)
    want = torch.zeros_like(pos)
    want[:S] = g.float().view(B, S, D).sum(0)
    assert torch.allclose(pos.grad, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16, torch.float16])
def test_label_fuse_matches_torch(dev, dt):
    """ops.label_fuse: (fuse + match_score . label_embs^T) * vmask (reference models/SeqPAN.py:80-82), forward and the
    three gradients (residual, probabilities, label embeddings) against torch fp32."""
    ops = _ops()
    M, K, N = 200, 256, 4
    torch.manual_seed(11)
    res = torch.randn(M, K, device=dev).to(dt).requires_grad_(True)
    p = torch.softmax(torch.randn(M, N, device=dev), -1).requires_grad_(True)
    E = torch.randn(K, N, device=dev, requires_grad=True)
    rs = (torch.rand(M, device=dev) > 0.3).float()
    y = ops.label_fuse(res, p, E, rs)
    r32, p32, E32 = res.detach().float().requires_grad_(True), p.detach().clone().requires_grad_(True), E.detach().clone().requires_grad_(True)
    ref = (r32 + p32 @ E32.t()) * rs[:, None]
    tol = 1e-5 if dt == torch.float32 else 2e-2
    _close(y.float(), ref, tol, "label_fuse fwd")
    g = torch.randn(M, K, device=dev).to(dt)
    y.backward(g)
    ref.backward(g.float())
    _close(res.grad.float(), r32.grad, tol, "dres")
    _close(p.grad, p32.grad, tol, "dprobs")
    _close(E.grad, E32.grad, tol, "dlabel_embs")


def test_transpose_batched_exact(dev):
    """vmr_transpose_batched (the K-major weight copies of the arena): the 16-byte tile path (whole 64 x 64 tiles) and the
    4-byte / element fallback, several items in one launch."""
    from vmrframe_amd import _lib as L
    shapes = [(128, 192), (64, 64), (100, 50), (1024, 3072), (70, 64), (256, 8)]
    srcs = [torch.randn(r, c, device=dev).to(torch.bfloat16) for r, c in shapes]
    dsts = [torch.zeros(c, r, device=dev, dtype=torch.bfloat16) for r, c in shapes]
    items = (L.TransposeItem * len(shapes))()
    for it, a, b, (r, c) in zip(items, srcs, dsts, shapes):
        it.src, it.dst, it.rows, it.cols = a.data_ptr(), b.data_ptr(), r, c
    L.check(L.lib().vmr_transpose_batched(items, len(shapes), L.stream_ptr()), "vmr_transpose_batched")
    for a, b, shp in zip(srcs, dsts, shapes):
        assert torch.equal(b, a.t().contiguous()), shp


def test_cq_score_row_split_equals_single_launch(dev):
    """vmr_cq_score_fwd_ws: a clip's long rows over 4 workgroups + the column-normalisation pass against the one-launch
    form -- the softmax over the short index bit-equal, the softmax over the long index to fp32 rounding; ragged
    lengths and fully masked row blocks (padded frames at the end of a clip) included."""
    from vmrframe_amd import _lib as L
    lib = L.lib()
    B, Ll, Ls, D = 6, 128, 20, 256
    torch.manual_seed(5)
    lng = (torch.randn(B, Ll, D, device=dev) / 4).to(torch.bfloat16)
    sht = (torch.randn(B, Ls, D, device=dev) / 4).to(torch.bfloat16)
    st = torch.randn(B, Ls, device=dev)
    ml = torch.ones(B, Ll, device=dev)
    ms = torch.ones(B, Ls, device=dev)
    ml[1, 70:] = 0; ml[2, 20:] = 0; ms[3, 11:] = 0; ml[4, 96:] = 0
    SP = (Ls + 7) // 8 * 8
    outs = []
    try:
        for mode in (0, 1):
            lib.vmr_debug_set_cq_split(mode)
            Pt = torch.empty(B, Ll, SP, device=dev); Pv = torch.empty_like(Pt)
            ws = torch.empty(lib.vmr_cq_score_ws_floats(B), device=dev)
            L.check(lib.vmr_cq_score_fwd_ws(lng.data_ptr(), sht.data_ptr(), st.data_ptr(), ml.data_ptr(), ms.data_ptr(), None, None,
                                            Pt.data_ptr(), Pv.data_ptr(), ws.data_ptr(), B, Ll, Ls, D, 0, 0, L.BF16, L.stream_ptr()),
                    "vmr_cq_score_fwd_ws")
            outs.append((Pt, Pv))
    finally:
        lib.vmr_debug_set_cq_split(-1)
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.allclose(outs[0][1], outs[1][1], rtol=2e-6, atol=1e-9)
    assert float(outs[1][1][:, :, :Ls].sum(1).sub(1).abs().max()) < 1e-5      # columns still sum to one over the long index
