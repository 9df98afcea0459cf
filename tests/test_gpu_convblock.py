"""GPU: the fused conv-block backward (csrc/convblock.hip, ops.conv_block) -- reference DepthwiseSeparableConvBlock,
models/layers.py:126-148.

(a) kernel level, through the C ABI: vmr_convblock_bwd against the three launches it replaces (vmr_dwconv_bwd2 +
    vmr_layernorm_bwd + the lower layer's vmr_relu_bwd_bias mode 3) on the same operands: dx and dz identical (the
    arithmetic and the rounding points are the same), parameter gradients to summation-order noise;
(b) op level: ops.conv_block (one autograd node, arena gradients) against a plain PyTorch fp32 restatement of the
    block with the kernel's own dropout masks;
(c) model level: the whole SeqPAN step with the one-node block on and off from identical state -- same logits, same
    gradient arena (to bf16 summation noise)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("D,segs", [(1024, ((3, 128), (3, 20))), (512, ((2, 64), (2, 10))), (1024, ((2, 37), (0, 0))),
                                    (512, ((1, 1), (2, 3))), (1024, ((1, 256), (2, 9)))])
def test_convblock_bwd_equals_the_three_kernels(dev, dt, D, segs):
    from vmrframe_amd import _lib as L
    lib = L.lib()
    st = L.stream_ptr()
    torch.manual_seed(5)
    (B1, S1), (B2, S2) = segs
    rows = B1 * S1 + B2 * S2
    code = L.dtype_code(torch.empty(0, dtype=dt))
    x = torch.randn(rows, D, device=dev).to(dt)
    du = torch.randn(rows, D, device=dev).to(dt)
    dres = torch.randn(rows, D, device=dev).to(dt)
    gamma = 1 + 0.1 * torch.randn(D, device=dev)
    beta = 0.1 * torch.randn(D, device=dev)
    w = torch.randn(D, 7, device=dev) / math.sqrt(7)
    bits = torch.randint(0, 256, (rows, D // 8), device=dev, dtype=torch.uint8)
    xf = x.float()
    mean = xf.mean(1)
    rstd = torch.rsqrt(xf.var(1, unbiased=False) + 1e-6)
    scale = 1.25
    # --- the three separate launches
    dn = torch.empty_like(x)
    dw_ref = torch.zeros(D, 7, device=dev)
    ws = torch.empty((B1 * ((S1 + 63) // 64) + B2 * ((S2 + 63) // 64)) * D * 7 + 16, device=dev)
    L.check(lib.vmr_dwconv_bwd2(du.data_ptr(), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                w.data_ptr(), dn.data_ptr(), dw_ref.data_ptr(), ws.data_ptr(), B1, S1, B2, S2, D, code, st), "dwconv_bwd2")
    dx_ref = torch.empty_like(x)
    dg_ref, db_ref = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    ws2 = torch.empty(L.ln_bwd_ws_floats(rows, D), device=dev)
    L.check(lib.vmr_layernorm_bwd(dn.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dres.data_ptr(),
                                  dx_ref.data_ptr(), dg_ref.data_ptr(), db_ref.data_ptr(), None, ws2.data_ptr(), 0, rows, D, code,
                                  0.0, 0, None, st), "layernorm_bwd")
    dz_ref = torch.empty_like(x)
    L.check(lib.vmr_relu_bwd_bias(3, dx_ref.data_ptr(), bits.data_ptr(), dz_ref.data_ptr(), None, rows, D, D, scale, code, 0.0, 0,
                                  None, None, 1.0, st), "relu_bwd_bias")
    # --- the fused launch (+ its column reductions)
    nbmax = lib.vmr_convblock_bwd_blocks(B1, S1, B2, S2, D)
    assert 0 < nbmax <= min(1024, B1 * ((S1 + 7) // 8) + B2 * ((S2 + 7) // 8))
    part_dw = torch.full((nbmax, 7 * D), float("nan"), device=dev)
    part_gb = torch.full((nbmax, 2 * D), float("nan"), device=dev)
    dx = torch.full_like(x, float("nan"))
    dz = torch.full_like(x, float("nan"))
    nb = C.c_int32(0)
    L.check(lib.vmr_convblock_bwd(du.data_ptr(), x.data_ptr(), dres.data_ptr(), bits.data_ptr(), scale, gamma.data_ptr(),
                                  beta.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), dx.data_ptr(), dz.data_ptr(),
                                  part_dw.data_ptr(), part_gb.data_ptr(), B1, S1, B2, S2, D, code, C.byref(nb), st), "convblock_bwd")
    assert nb.value == nbmax
    torch.cuda.synchronize()
    assert torch.isfinite(dx.float()).all() and torch.isfinite(dz.float()).all() and torch.isfinite(part_dw).all()
    # same arithmetic, same rounding points: identical up to the compiler's choice of fused multiply-adds
    tol = 1e-5 if dt == torch.float32 else 4e-3
    assert rel(dx.float(), dx_ref.float()) < tol
    mism = (dx != dx_ref)
    if dt != torch.float32:    # where dx agrees bit for bit, dz does too
        assert float(mism.float().mean()) < 2e-2
        assert torch.equal(dz[~mism], dz_ref[~mism])
    assert rel(dz.float(), dz_ref.float()) < tol
    dw = part_dw.sum(0).view(D, 7)
    gb = part_gb.sum(0)
    assert rel(dw, dw_ref) < 2e-4 and rel(gb[:D], dg_ref) < 2e-4 and rel(gb[D:], db_ref) < 2e-4
    # without a lower layer: dx only, nothing touches dz
    dx2 = torch.empty_like(x)
    L.check(lib.vmr_convblock_bwd(du.data_ptr(), x.data_ptr(), dres.data_ptr(), None, 1.0, gamma.data_ptr(), beta.data_ptr(),
                                  mean.data_ptr(), rstd.data_ptr(), w.data_ptr(), dx2.data_ptr(), None, part_dw.data_ptr(),
                                  part_gb.data_ptr(), B1, S1, B2, S2, D, code, C.byref(nb), st), "convblock_bwd")
    assert torch.equal(dx2, dx)


def test_convblock_bwd_rejects_unsupported_widths(dev):
    from vmrframe_amd import _lib as L
    lib = L.lib()
    assert lib.vmr_convblock_bwd_supported(1024, L.BF16) and lib.vmr_convblock_bwd_supported(512, L.F32)
    assert not lib.vmr_convblock_bwd_supported(64, L.BF16) and not lib.vmr_convblock_bwd_supported(2048, L.BF16)
    t = torch.zeros(8, 64, device=dev, dtype=torch.bfloat16)
    f = torch.zeros(8 * 64 * 8, device=dev)
    nb = C.c_int32(0)
    rc = lib.vmr_convblock_bwd(t.data_ptr(), t.data_ptr(), t.data_ptr(), None, 1.0, f.data_ptr(), f.data_ptr(), f.data_ptr(),
                               f.data_ptr(), f.data_ptr(), t.data_ptr(), None, f.data_ptr(), f.data_ptr(), 1, 8, 0, 0, 64, L.BF16,
                               C.byref(nb), L.stream_ptr())
    assert rc != 0 and b"not supported" in lib.vmr_last_error()


def _seqpan(dev, compute, droprate=0.0, train=False):
    """The cfg1 model (T=64, L=10, D=512, weights by the golden recipe) on a 64-clip synthetic batch: 64 x 74 packed
    tokens = 37 x 128 rows, so the conv-block products run on the LDS-DMA kernels whose epilogue writes the bit masks
    (at the fixture's own B = 4 they take the register-staged kernel and the block stays on separate nodes)."""
    from oracle import seqpan_ref as R
    from tests.test_gpu_trainer import build
    z, cfg, _, _, weights = load_golden("g_cfg1")
    B, T, L, D, V, nw, nc, C, seed = [int(v) for v in z["meta"]]
    B = 64
    batch = R.synth_batch(B, T, L, V, nw, nc, C=C, seed=5)
    g = R.gumbel_noise(B, T, 5)
    model = build(cfg, weights, compute, dev, g, droprate=droprate, train=train)
    return z, cfg, batch, model


@pytest.mark.parametrize("compute,drop", [("bf16", 0.0), ("bf16", 0.2), ("fp16", 0.0)])
def test_model_step_with_and_without_the_one_node_block(dev, compute, drop):
    """Same model state, same batch, same dropout seeds: the step with ops.conv_block (one node per block, fused
    backward) against the step with ln_dwconv + linear nodes.  Forward kernels are the same launches -> identical
    logits; gradients agree to summation-order noise of the parameter-gradient reductions."""
    import vmrframe_amd as V
    from vmrframe_amd import ops
    from vmrframe_amd.optim import FlatAdamW
    z, cfg, batch, model = _seqpan(dev, compute, droprate=drop, train=drop > 0)
    opt = FlatAdamW(model, lr=0.0, weight_decay=0.0, max_norm=1.0)
    res = {}
    calls = {"n": 0}
    orig = ops.conv_block

    def counting(*a, **k):
        calls["n"] += 1
        return orig(*a, **k)
    ops.conv_block = counting
    try:
        opt.zero_grad()                  # warm-up step: the first optimizer step builds the flat arenas (lr = 0)
        loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
        opt.backward(loss)
        opt.step()
        assert calls["n"] == 0           # (no arena yet: the block ran as separate nodes)
        for fused in (False, True, False):
            ops.FUSED_CONV_BLOCK = fused
            opt.zero_grad()
            model._seed_calls = 10       # same dropout sites / seeds in every run
            loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
            opt.backward(loss)
            torch.cuda.synchronize()
            res.setdefault(fused, []).append((loss.item(), out["slogits"].float().clone(), out["elogits"].float().clone(),
                                              opt.arena.flat_g.clone()))
    finally:
        ops.conv_block = orig
        ops.FUSED_CONV_BLOCK = True
    assert calls["n"] == 3, "the one-node conv block did not run for the encoder and both predictor passes"
    (l0, s0, e0, g0), (l2, s2, e2, g2) = res[False]
    (l1, s1, e1, g1) = res[True][0]
    assert torch.equal(s0, s1) and torch.equal(e0, e1) and abs(l0 - l1) <= 1e-6 * abs(l0)   # same forward launches
    noise = rel(g2, g0)                                                    # run-to-run noise of the unfused path (atomics)
    err = rel(g1, g0)
    # per tensor: the fused kernel computes dn with packed FMAs (v_pk_fma_f32), so ~1 % of the bf16 dn / dx values land
    # one ulp away from the unfused kernels' and the difference travels down the backward pass; the lower layers' bias
    # gradients are summed from the rounded dz (MFMA column sums) instead of the unrounded product
    named = dict(model.named_parameters())
    per = []
    for n in opt.names:
        o, k = opt.offsets[n], named[n].numel()
        ref = g0[o:o + k]
        if float(ref.norm()) >= 1e-3 * float(g0.norm()):
            per.append((rel(g1[o:o + k], ref), n))
    worst = max(per)
    conv = max(e for e in per if "conv_block" in e[1])
    print(f"[convblock {compute} drop {drop}] fused vs unfused gradient arena: {err:.3e} (unfused run-to-run {noise:.3e}); "
          f"worst tensor {worst}, worst conv-block tensor {conv}")
    assert err < 1e-2 and worst[0] < 3e-2, (err, worst)


def test_conv_block_matches_torch_reference(dev):
    """ops.conv_block on arena-backed parameters against a PyTorch fp32 restatement of the reference block (eval
    mode: no dropout), forward and every gradient."""
    from vmrframe_amd import ops
    from vmrframe_amd.optim import FlatArena
    torch.manual_seed(11)
    D, segs, nl = 512, ((4, 56), (4, 8)), 3        # 256 packed rows: products on the LDS-DMA kernels (bit-mask epilogue)
    rows = sum(b * s for b, s in segs)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.g = torch.nn.ParameterList([torch.nn.Parameter(1 + 0.1 * torch.randn(D)) for _ in range(nl)])
            self.b = torch.nn.ParameterList([torch.nn.Parameter(0.1 * torch.randn(D)) for _ in range(nl)])
            self.dw = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(D, 1, 7) / math.sqrt(7)) for _ in range(nl)])
            self.pw = torch.nn.ParameterList([torch.nn.Parameter(torch.randn(D, D, 1) / math.sqrt(D)) for _ in range(nl)])
            self.pb = torch.nn.ParameterList([torch.nn.Parameter(0.1 * torch.randn(D)) for _ in range(nl)])
    m = M().to(dev)
    for p in m.parameters():
        p.grad = torch.zeros_like(p)
    arena = FlatArena(m, mirror_dtype=torch.bfloat16)
    arena.refresh_transposed()
    cache = ops.WeightCache()
    layers = [(m.g[l], m.b[l], m.dw[l], m.pw[l], m.pb[l]) for l in range(nl)]
    x = torch.randn(rows, D, device=dev).to(torch.bfloat16).requires_grad_(True)
    assert ops.conv_block_fusable(x, layers)
    y = ops.conv_block(x, cache, segs, 1e-6, [ops.NO_DROP] * nl, layers)
    gy = torch.randn(rows, D, device=dev).to(torch.bfloat16)
    arena.flat_g.zero_()
    (gx,) = torch.autograd.grad(y, [x], gy)
    cache.state.flush_reduce(); cache.state.flush_colreduce()
    torch.cuda.synchronize()
    # reference
    ps = [p.detach().clone().requires_grad_(True) for p in m.parameters()]
    names = [n for n, _ in m.named_parameters()]
    P = dict(zip(names, ps))
    xr = x.detach().float().requires_grad_(True)
    h = xr
    for l in range(nl):
        outs, r = [], 0
        for (B, S) in segs:
            n = torch.nn.functional.layer_norm(h[r:r + B * S].view(B, S, D), (D,), P[f"g.{l}"], P[f"b.{l}"], 1e-6)
            c = torch.nn.functional.conv1d(n.transpose(1, 2), P[f"dw.{l}"], padding=3, groups=D).transpose(1, 2)
            outs.append(c.reshape(B * S, D)); r += B * S
        u = torch.cat(outs, 0)
        wq = P[f"pw.{l}"][:, :, 0]
        h = torch.relu(u @ (wq.detach().to(torch.bfloat16).float() + (wq - wq.detach())).t() + P[f"pb.{l}"]) + h
    assert rel(y.float(), h) < 2e-2
    grads = torch.autograd.grad(h, [xr] + ps, gy.float())
    # (three bf16 layers against an fp32 reference; the tight comparison -- against the separate-node path on the same
    #  operands -- is the model-level test above)
    assert rel(gx.float(), grads[0]) < 6e-2
    for (n, p), gr in zip(m.named_parameters(), grads[1:]):
        mg = ops.main_grad(p)
        assert mg is not None
        assert rel(mg.view_as(gr), gr) < 6e-2, n


@pytest.mark.parametrize("dt,tol", [(torch.float32, 3e-5), (torch.bfloat16, 3e-2), (torch.float16, 3e-3)])
@pytest.mark.parametrize("D,segs", [(512, [(3, 16), (3, 6)]), (1024, [(2, 128), (2, 20)]), (512, [(1, 1), (2, 3)]), (1024, [(2, 70)]),
                                    (1024, [(5, 37), (4, 9)])])
def test_ln_dwconv_forward_persistent_kernel_matches_torch(dev, dt, tol, D, segs):
    """vmr_ln_dwconv_fwd2 at D = 512 / 1024 runs convblock.hip's persistent forward kernel (norm.hip's two-phase tile
    kernel keeps the other widths, tests/test_gpu_a_ops.py): u, mean, rstd against PyTorch, ragged sequence groups."""
    from vmrframe_amd import _lib as L
    lib = L.lib()
    torch.manual_seed(2)
    rows = sum(b * s for b, s in segs)
    x = (torch.randn(rows, D, device=dev) * 1.5 + 0.3).to(dt)
    gamma = 1 + 0.1 * torch.randn(D, device=dev)
    beta = 0.1 * torch.randn(D, device=dev)
    w = torch.randn(D, 7, device=dev) / math.sqrt(7)
    u = torch.full_like(x, float("nan"))
    mean = torch.full((rows,), float("nan"), device=dev)
    rstd = torch.full((rows,), float("nan"), device=dev)
    (B1, S1), (B2, S2) = segs[0], (segs[1] if len(segs) > 1 else (0, 0))
    L.check(lib.vmr_ln_dwconv_fwd2(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-6, w.data_ptr(), u.data_ptr(), mean.data_ptr(),
                                   rstd.data_ptr(), B1, S1, B2, S2, D, L.dtype_code(x), L.stream_ptr()), "vmr_ln_dwconv_fwd2")
    torch.cuda.synchronize()
    xf = x.float()
    assert torch.allclose(mean, xf.mean(1), atol=1e-5, rtol=1e-5)
    assert torch.allclose(rstd, torch.rsqrt(xf.var(1, unbiased=False) + 1e-6), atol=1e-5, rtol=1e-4)
    outs, r = [], 0
    for (B, S) in segs:
        n = torch.nn.functional.layer_norm(xf[r:r + B * S].view(B, S, D), (D,), gamma, beta, 1e-6)
        n = n.to(dt).float()                                   # the kernel stages LN(x) in the storage type
        c = torch.nn.functional.conv1d(n.transpose(1, 2), w.view(D, 1, 7), padding=3, groups=D).transpose(1, 2)
        outs.append(c.reshape(B * S, D)); r += B * S
    ref = torch.cat(outs, 0)
    assert torch.isfinite(u.float()).all()
    assert rel(u.float(), ref) < tol
