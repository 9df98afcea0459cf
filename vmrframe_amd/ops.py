"""Autograd operators of the SeqPAN path, each a thin wrapper that launches the
hand-written HIP kernels of libvmr_hip.so through the C ABI (include/vmr_hip.h)
on torch's current HIP stream.  PyTorch supplies device memory, streams and the
autograd tape only; there is no non-HIP fallback (see _lib.require_gpu).

Dropout never stores a mask: every site gets a (p, seed, step_ptr) triple from
`DropCtx`; forward and backward kernels regenerate the same counter-based mask.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib as L

NO_DROP = (0.0, 0, None)
GEMM_HOOK = None   # bench.py: callable(launch, M, N, K, ta, tb, Z, dtype) timing the launch with HIP events
GEMM2_HOOK = None  # bench.py: callable(launch, flops) around the merged dX + dW (+ slab reduction) launches
CQ_HOOK = None     # bench.py: callable(launch, B, Ll, Ls, D) around the CQAttention score kernel
# Optional side stream for the weight-gradient GEMMs (dW = dY^T.X).  They are off the backward's
# critical path (only the optimizer needs them), so the trainer lets them run beside the dX chain:
# the hardware fills the single-round tails / epilogue bursts of one kernel with workgroups of the
# other instead of idling at every in-order kernel boundary.  Whoever sets this MUST make the main
# stream wait for it after backward() (trainer.GraphedTrainStep does); default None = in-order.
DW_SIDE_STREAM = None


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _cdiv(x: int, m: int) -> int:
    return (x + m - 1) // m


def _rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class DropCtx:
    """Per-forward dropout bookkeeping: hands out one seed per call site."""

    def __init__(self, p: float, training: bool, base_seed: int, step: Optional[torch.Tensor] = None):
        self.p = float(p) if training else 0.0
        self.base = base_seed & 0xFFFFFFFF
        self.n = 0
        self.step = step  # optional device int32[1] mixed into every seed (hipGraph replay)
        self.sites: List[Tuple[str, int]] = []

    def next(self, name: str = "") -> Tuple[float, int, Optional[torch.Tensor]]:
        if self.p <= 0.0:
            return NO_DROP
        self.n += 1
        seed = (self.base * 0x9E3779B1 + self.n * 0x85EBCA6B + 0x1234567) & 0xFFFFFFFF
        self.sites.append((name, seed))
        return (self.p, seed, self.step)

    def noise_seed(self, name: str = "") -> Tuple[int, Optional[torch.Tensor]]:
        """Seed of a sampling site that is active in eval mode too (the Gumbel noise of the match head)."""
        self.n += 1
        seed = (self.base * 0x9E3779B1 + self.n * 0x85EBCA6B + 0x7654321) & 0xFFFFFFFF
        self.sites.append((name, seed))
        return seed, self.step


# ---------------------------------------------------------------------------
# raw GEMM launch
# ---------------------------------------------------------------------------
class HeldGemm:
    """A product whose launch is held back by its caller so that a following split-K slab product (the same layer's
    weight gradient) can take it along in ONE launch (vmr_gemm2).  Purely local to one `_Linear.backward` call."""
    __slots__ = ("desc", "keep", "flops", "nbytes")

    def __init__(self, desc, keep, flops, nbytes):
        self.desc, self.keep, self.flops, self.nbytes = desc, keep, flops, nbytes

    def launch(self):
        L.check(L.lib().vmr_gemm(C.byref(self.desc), L.stream_ptr()), "vmr_gemm")


class PassState:
    """Launches that ONE model's backward pass has held back: the split-K second stage that rides in the next layer's
    merged launch, and the parameter-gradient column reductions that run as one batched launch when the autograd
    engine finishes the pass.  One instance per model (it hangs off the model's WeightCache), so two models -- or a
    model whose previous backward died half-way -- never see each other's partials."""

    def __init__(self):
        self.pending_reduce = None   # (slabs kept alive, dst, nsplit, n, cols, ld_dst)
        self.pending_stream = None   # the stream the pending slabs were produced on
        self.deferred: List[tuple] = []   # (partials kept alive, out0, out1, nblocks, n0, n1, slots)
        self.deferred_streams: List = []  # streams the queued partial rows were produced on

    def reset(self):
        """Drop (never launch) whatever a dead backward pass left behind: stale partials must not be added into the
        freshly zeroed gradient arena.  Called at every forward and from FlatAdamW.zero_grad()."""
        self.pending_reduce = None
        self.pending_stream = None
        self.deferred.clear()
        self.deferred_streams.clear()

    @staticmethod
    def _after(stream):
        """The current stream waits for `stream` (a held-back second stage may be launched from another stream than
        the one that produced its partials: two branches of the backward pass on two streams)."""
        if stream is not None and torch.cuda.is_available():
            cur = torch.cuda.current_stream()
            if cur != stream:
                cur.wait_stream(stream)

    # -- split-K second stage ---------------------------------------------------
    def take_reduce(self):
        rj, self.pending_reduce = self.pending_reduce, None
        if rj is not None:
            self._after(self.pending_stream)
        self.pending_stream = None
        return rj

    def flush_reduce(self):
        rj = self.take_reduce()
        if rj is not None:
            ws_, dst_, sk_, n_, cols_, ld_ = rj
            L.check(L.lib().vmr_splitk_reduce(ws_.data_ptr(), dst_.data_ptr(), sk_, n_, cols_, ld_, L.stream_ptr()),
                    "vmr_splitk_reduce")

    def reduce_later(self, ws, dst, sk, n, cols, ld):
        """Second stage of a split-K weight gradient: held back so that the NEXT merged dX + dW launch (the next layer
        of the backward pass) carries it as extra workgroups; whatever is still pending when the autograd engine
        finishes the pass is launched then (queue_callback), i.e. before anything can read the gradient arena."""
        if not DEFER_SPLITK_REDUCE or DW_SIDE_STREAM is not None or not MERGE_DX_DW:
            L.check(L.lib().vmr_splitk_reduce(ws.data_ptr(), dst.data_ptr(), sk, n, cols, ld, L.stream_ptr()), "vmr_splitk_reduce")
            return
        self.flush_reduce()
        try:   # (every time, see defer_colreduce; the callback is idempotent)
            torch.autograd.Variable._execution_engine.queue_callback(self.flush_reduce)
        except RuntimeError:      # not inside a backward pass
            L.check(L.lib().vmr_splitk_reduce(ws.data_ptr(), dst.data_ptr(), sk, n, cols, ld, L.stream_ptr()),
                    "vmr_splitk_reduce")
            return
        self.pending_reduce = (ws, dst, sk, n, cols, ld)
        self.pending_stream = torch.cuda.current_stream() if ws.is_cuda else None

    # -- parameter-gradient column reductions -------------------------------------
    def defer_colreduce(self, part, out0, out1, nblocks, n0, n1, slots):
        """Queue the second stage of a parameter-gradient reduction; all queued items of one backward pass run as ONE
        launch when the autograd engine finishes the pass (queue_callback), before anything can read the arena."""
        # (queued on every call, not only the first of a pass: the flush is idempotent)
        torch.autograd.Variable._execution_engine.queue_callback(self.flush_colreduce)
        self.deferred.append((part, out0, out1, int(nblocks), int(n0), int(n1), int(slots)))
        if part.is_cuda:
            st = torch.cuda.current_stream()
            if st not in self.deferred_streams:
                self.deferred_streams.append(st)

    def flush_colreduce(self):
        if not self.deferred:
            return
        items = (L.ColReduceItem * len(self.deferred))()
        for it, (part, o0, o1, nb, n0, n1, sl) in zip(items, self.deferred):
            it.part, it.out0, it.out1 = part.data_ptr(), o0.data_ptr(), o1.data_ptr()
            it.nblocks, it.n0, it.n1, it.slots = nb, n0, n1, sl
        n = len(self.deferred)
        for st in self.deferred_streams:
            self._after(st)
        try:
            L.check(L.lib().vmr_colreduce_batched(items, n, L.stream_ptr()), "vmr_colreduce_batched")
        finally:
            self.deferred.clear()
            self.deferred_streams.clear()


def gemm(A, B, Cmat, M, N, K, ta, tb, lda, ldb, ldc, *, dtype, flags=0, bias=None, residual=None, aux=None,
         ldr=0, rowscale=None, alpha=1.0, Z1=1, Z2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), splitk=1, drop=NO_DROP,
         bias2=None, bias_scale=1.0, res_div=1, a_colsum=None, hold=False, held: Optional[HeldGemm] = None,
         state: Optional[PassState] = None):
    """One vmr_gemm launch.  hold=True: do not launch, return a HeldGemm.  held=<HeldGemm>: launch the held product
    too -- in ONE launch with this one when this is a split-K slab product (vmr_gemm2; `state`'s pending slab
    reduction then rides along as well), otherwise on its own just before this one."""
    d = L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), Cmat.data_ptr()
    d.bias, d.residual, d.aux, d.rowscale = _ptr(bias), _ptr(residual), _ptr(aux), _ptr(rowscale)
    d.lda, d.ldb, d.ldc, d.ldr = lda, ldb, ldc, ldr
    d.M, d.N, d.K, d.transA, d.transB = M, N, K, ta, tb
    d.dtype, d.flags, d.alpha = dtype, flags, alpha
    d.Z1, d.Z2 = Z1, Z2
    d.sA1, d.sA2 = sA
    d.sB1, d.sB2 = sB
    d.sC1, d.sC2 = sC
    d.splitk = splitk
    d.drop_p, d.drop_seed = drop[0], drop[1]
    d.drop_step = _ptr(drop[2])
    d.bias2, d.bias_scale, d.res_div = _ptr(bias2), bias_scale, res_div
    d.a_colsum = _ptr(a_colsum)

    def single(desc=d):
        L.check(L.lib().vmr_gemm(C.byref(desc), L.stream_ptr()), "vmr_gemm")

    if hold:
        return HeldGemm(d, (A, B, Cmat, bias, residual, aux, rowscale, bias2, a_colsum, drop),
                        2.0 * M * N * K, 2.0 * (M * K + N * K + M * N))
    if held is not None and not (ta and tb and splitk > 1 and (flags & L.EPI_SLAB) and Z1 * Z2 == 1):
        held.launch()           # not a slab product: nothing to share, the held product goes first on its own
        held = None
    if held is not None:
        pd = held.desc
        rj = state.take_reduce() if state is not None else None

        def merged():
            if rj is None:
                L.check(L.lib().vmr_gemm2(C.byref(pd), C.byref(d), L.stream_ptr()), "vmr_gemm2")
            else:   # the previous layer's slab reduction rides in this launch
                ws_, dst_, sk_, n_, cols_, ld_ = rj
                L.check(L.lib().vmr_gemm2_reduce(C.byref(pd), C.byref(d), ws_.data_ptr(), dst_.data_ptr(), sk_, n_, cols_,
                                                 ld_, L.stream_ptr()), "vmr_gemm2_reduce")
        if GEMM2_HOOK is not None:
            # algorithmic bytes (the minimum any schedule must move): dX operands + result, dW operands + ONE fp32
            # [M, N] result.  The split-K slabs and the ridden reduction's read-modify-write are this
            # implementation's overhead and show up in `traffic`, not here.
            GEMM2_HOOK(merged, held.flops + 2.0 * M * N * K, held.nbytes + 2.0 * (M * K + N * K) + 4.0 * M * N)
        else:
            merged()
        return None
    if GEMM_HOOK is not None:
        GEMM_HOOK(single, M, N, K, ta, tb, Z1 * Z2, dtype)
        return None
    single()
    return None


MERGE_DX_DW = os.environ.get("VMR_MERGE_DX_DW", "1") != "0"
CQ_STREAMS = os.environ.get("VMR_CQ_STREAMS", "0") != "0"   # experiment: the v2q CQAttention direction on a side stream
CQ_TEE = os.environ.get("VMR_CQ_TEE", "1") != "0"          # dropout backward + other-consumer gradient in one kernel; one-concat row split
AUX_BITS = os.environ.get("VMR_AUX_BITS", "1") != "0"      # ReLU / dropout masks of the conv-block products as bit matrices
GROUP_DW = os.environ.get("VMR_GROUP_DW", "1") != "0"      # one weight-gradient product per grouped projection
DEFER_SPLITK_REDUCE = os.environ.get("VMR_DEFER_SPLITK_REDUCE", "1") != "0"


def mm(a: torch.Tensor, b: torch.Tensor, ta: int, tb: int, *, out=None, out_f32=False, **kw) -> torch.Tensor:
    """2-D product through vmr_gemm; a/b may be row-strided views (last stride 1)."""
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    M, K = (a.shape[1], a.shape[0]) if ta else a.shape
    N = b.shape[1] if tb else b.shape[0]
    assert (b.shape[0] if tb else b.shape[1]) == K, (a.shape, b.shape, ta, tb)
    dt = L.dtype_code(a)
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    flags = kw.pop("flags", 0) | (L.EPI_OUT_F32 if (out_f32 and dt != L.F32) else 0)
    h = gemm(a, b, out, M, N, K, ta, tb, a.stride(0), b.stride(0), out.stride(0), dtype=dt, flags=flags, **kw)
    return (out, h) if kw.get("hold") else out


def mm_few_tiles(a: torch.Tensor, b: torch.Tensor, ta: int, tb: int, bias=None, addend=None) -> Optional[torch.Tensor]:
    """a . b (+ bias[n] + addend[m,n]) in the compute dtype for products with few output tiles and a long K (a [64, D]
    pooled vector through a [D, D] weight; the text-side [1280, 1024, 4096] product): K is split over the idle CUs into
    plain fp32 slabs and ONE second-stage launch sums them, applies the epilogue and casts (no zero-fill, no atomics:
    the summation order is fixed).  None if the shape does not qualify (the caller uses the plain path)."""
    M, K = (a.shape[1], a.shape[0]) if ta else a.shape
    N = b.shape[1] if tb else b.shape[0]
    tiles = _cdiv(M, 128) * _cdiv(N, 128)
    if a.dtype == torch.float32 or tiles >= FEW_TILES or K < 512 or N % 4 != 0:
        return None
    # enough splits for ~320 workgroups, at least two 128-deep K-steps each (the text-side products: [1280,1024,4096]
    # is 80 tiles x 64 K-steps as a plain launch -- 63 us on a third of the CUs)
    sk = max(2, min(8, K // 256 if tiles >= 32 else K // 128, _cdiv(320, tiles)))
    ws = torch.empty(sk, M, N, device=a.device, dtype=torch.float32)
    gemm(a, b, ws, M, N, K, ta, tb, a.stride(0), b.stride(0), N, dtype=L.dtype_code(a), flags=L.EPI_SLAB, splitk=sk)
    out = torch.empty(M, N, device=a.device, dtype=a.dtype)
    if addend is not None:
        addend = addend.contiguous()
        assert tuple(addend.shape) == (M, N) and addend.dtype == a.dtype
    L.check(L.lib().vmr_splitk_reduce_cast(ws.data_ptr(), sk, M, N, _ptr(bias), _ptr(addend), out.data_ptr(), L.dtype_code(out),
                                           L.stream_ptr()), "vmr_splitk_reduce_cast")
    return out


def bmm4(a: torch.Tensor, b: torch.Tensor, c: torch.Tensor, ta: int, tb: int, *, flags=0, **kw):
    """Batched product over two leading dims of 4-D strided views:
    a [Z1,Z2,M,K] (or [..,K,M] if ta), b [Z1,Z2,N,K] (or [..,K,N] if tb), c [Z1,Z2,M,N]."""
    assert a.stride(3) == 1 and b.stride(3) == 1 and c.stride(3) == 1
    Z1, Z2 = c.shape[0], c.shape[1]
    M, N = c.shape[2], c.shape[3]
    K = a.shape[2] if ta else a.shape[3]
    dt = L.dtype_code(a)
    if c.dtype == torch.float32 and dt != L.F32:
        flags |= L.EPI_OUT_F32
    gemm(a, b, c, M, N, K, ta, tb, a.stride(2), b.stride(2), c.stride(2), dtype=dt, flags=flags,
         Z1=Z1, Z2=Z2, sA=(a.stride(0), a.stride(1)), sB=(b.stride(0), b.stride(1)),
         sC=(c.stride(0), c.stride(1)), **kw)


# workgroups to aim for in dW products.  512 when the product runs alone (8 splits for a 64-tile [1024 x 1024]
# gradient: two workgroups per CU); 384 when it shares its launch with the layer's single-round dX product (4 splits:
# half the slab traffic, the dX workgroups fill the other slots: 9.35 -> 9.27 ms/step at cfg2, but 8.7 -> 9.5 ms for
# BaseFast at T = 256, whose dX grids are multi-round and never merge)
FEW_TILES = int(os.environ.get("VMR_FEW_TILES", "128"))     # products below this many 128x128 tiles split K over the idle CUs
SPLITK_TARGET = int(os.environ.get("VMR_SPLITK_TARGET", "512"))
SPLITK_TARGET_MERGED = int(os.environ.get("VMR_SPLITK_TARGET_MERGED", "384"))
USE_SLABS = os.environ.get("VMR_SPLITK_SLABS", "1") != "0"
FUSED_ATTENTION = os.environ.get("VMR_FUSED_ATTN", "1") != "0"   # csrc/attention.hip forward (bf16, hd 128/256)
FUSED_ATTENTION_BWD = os.environ.get("VMR_FUSED_ATTN_BWD", "1") != "0"   # csrc/attention_bwd.hip
FUSED_CQ_SCORE = os.environ.get("VMR_FUSED_CQ", "1") != "0"   # csrc/cqscore.hip


def splitk_for(M: int, N: int, K: int, target: int = 0) -> int:
    """dW products have few output tiles and a long K (= tokens): split K until the grid has about
    one workgroup per CU; every split costs one more fp32 atomic pass over the [M,N] output."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    sk = 1
    target = target or SPLITK_TARGET
    while tiles * sk * 2 <= target and K // (sk * 2) >= 512:
        sk *= 2
    return sk


def main_grad(p):
    """fp32 gradient slot of a parameter inside the flat gradient arena (optim.FlatArena), if any:
    weight-gradient kernels then accumulate straight into it and autograd sees no gradient."""
    return getattr(p, "_vmr_main_grad", None)


# ---------------------------------------------------------------------------
# deferred parameter-gradient reductions (LayerNorm gamma/beta, depthwise-conv weights)
# ---------------------------------------------------------------------------
DEFER_COLREDUCE = os.environ.get("VMR_DEFER_COLREDUCE", "1") != "0"
USE_WT = os.environ.get("VMR_USE_WT", "1") != "0"     # dX products on the K-major weight copies


# ---------------------------------------------------------------------------
# compute-dtype weight cache
# ---------------------------------------------------------------------------
class WeightCache:
    """Compute-dtype copies of the fp32 master parameters (cast once per
    optimizer step; keyed on the parameters' version counters).  A group of
    weights that share their input (q/k/v, bilinear_1/2) is stored concatenated so
    one GEMM serves the group; K is zero-padded to a multiple of 8."""

    def __init__(self):
        self.store = {}
        self.state = PassState()     # the owning model's held-back backward launches

    def clear(self):
        """Forget the cast copies (the masters changed); mirror-backed entries stay valid because
        the optimizer kernel rewrites the mirror in place."""
        self.store = {k: v for k, v in self.store.items() if getattr(v[1], "_vmr_mirror", False)}

    @staticmethod
    def get_t(params: Sequence[torch.Tensor], dtype: torch.dtype) -> Optional[torch.Tensor]:
        """K-major copy [K, sum N] of the (grouped) weight, if the optimizer arena keeps one (optim.FlatArena:
        bf16, refreshed by one batched transpose per step); None otherwise."""
        if not L.is_16bit(dtype) or not USE_WT:
            return None
        return (getattr(params[0], "_vmr_wt_views", None) or {}).get(tuple(id(p) for p in params))

    def get(self, params: Sequence[torch.Tensor], dtype: torch.dtype, kpad: int = 0) -> torch.Tensor:
        """kpad > K: zero-pad the copy's K to kpad columns (an input padded to a multiple of 64 elements keeps
        its GEMM on the LDS-DMA kernel: V = 500 video features -> 512)."""
        key = (tuple(id(p) for p in params), dtype, kpad)
        ver = tuple(p._version for p in params) + tuple(p.data_ptr() for p in params)
        hit = self.store.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
        if L.is_16bit(dtype):
            # masters living in the flat arena have a 16-bit mirror (in the model's compute dtype) maintained by the AdamW kernel
            # (optim.FlatArena): a group laid out back to back is used in place, no cast launch
            mirrors = [getattr(p, "_vmr_w16", None) for p in params]
            if all(m is not None and m.dtype == dtype for m in mirrors):
                if any(p._version != getattr(p, "_vmr_synced_version", p._version) for p in params):
                    # a master was edited in place behind the optimizer's back (load_state_dict, p.copy_()): the
                    # mirror the AdamW kernel maintains -- and the K-major copies -- have not seen it yet
                    params[0]._vmr_arena.sync_mirrors()
                K = mirrors[0].numel() // mirrors[0].shape[0]
                ok = K % 8 == 0 and kpad in (0, K)
                for a_, b_ in zip(mirrors[:-1], mirrors[1:]):
                    ok = ok and a_.data_ptr() + a_.numel() * 2 == b_.data_ptr()
                if ok:
                    n = sum(m.shape[0] for m in mirrors)
                    w = torch.as_strided(mirrors[0], (n, K), (K, 1))
                    w._vmr_mirror = True
                    self.store[key] = (ver, w)
                    return w
        with torch.no_grad():
            mats = [p.detach().reshape(p.shape[0], -1).contiguous() for p in params]
            K = mats[0].shape[1]
            Kp = max(_rup(K, 8), kpad)
            N = sum(m.shape[0] for m in mats)
            if len(mats) == 1 and dtype == torch.float32 and Kp == K:
                w = mats[0]
            else:
                w = torch.empty(N, Kp, device=mats[0].device, dtype=dtype)
                r = 0
                for m in mats:
                    L.check(L.lib().vmr_cast(m.data_ptr(), L.F32, w[r:].data_ptr(), L.dtype_code(w), m.shape[0], K,
                                             m.stride(0), Kp, 0.0, 0, None, L.stream_ptr()), "vmr_cast")
                    r += m.shape[0]
        self.store[key] = (ver, w)
        return w


# ---------------------------------------------------------------------------
# cast (+pad, +input dropout)
# ---------------------------------------------------------------------------
def cast_pad(x: torch.Tensor, dtype: torch.dtype, drop=NO_DROP, mult: int = 8) -> torch.Tensor:
    """[rows, cols] fp32 -> [rows, roundup(cols, mult)] compute dtype, zero padded, with the
    VisualProjection input dropout (reference models/layers.py:120).  No gradient."""
    L.require_gpu(x)
    x = x.contiguous()
    rows, cols = x.shape
    out = torch.empty(rows, _rup(cols, mult), device=x.device, dtype=dtype)
    L.check(L.lib().vmr_cast(x.data_ptr(), L.dtype_code(x), out.data_ptr(), L.dtype_code(out), rows, cols, cols,
                             out.shape[1], drop[0], drop[1], _ptr(drop[2]), L.stream_ptr()), "vmr_cast")
    return out


class _ToDtype(torch.autograd.Function):
    """Differentiable dtype cast through vmr_cast (fp32 glue tensors <-> compute dtype);
    2-D inputs get their columns zero-padded to a multiple of 8 (16-byte GEMM rows)."""

    @staticmethod
    def forward(ctx, x, dtype, pad8):
        ctx.src = x.dtype
        ctx.shape = tuple(x.shape)
        x = x.contiguous()
        pad8 = pad8 and x.dim() == 2
        if pad8:
            rows, cols = x.shape
        else:
            rows, cols = 1, x.numel()
        ldd = _rup(cols, 8) if pad8 else cols
        if x.dtype == dtype and ldd == cols:
            return x
        out = torch.empty((rows, ldd) if pad8 else x.shape, device=x.device, dtype=dtype)
        L.check(L.lib().vmr_cast(x.data_ptr(), L.dtype_code(x), out.data_ptr(), L.dtype_code(out), rows, cols,
                                 cols, ldd, 0.0, 0, None, L.stream_ptr()), "vmr_cast")
        return out

    @staticmethod
    def backward(ctx, g):
        shape = ctx.shape
        if g.dtype == ctx.src and tuple(g.shape) == shape:
            return g, None, None
        g = g.contiguous()
        out = torch.empty(shape, device=g.device, dtype=ctx.src)
        if tuple(g.shape) != shape:
            rows, cols, lds = shape[0], shape[1], g.shape[1]
        else:
            rows, cols, lds = 1, g.numel(), g.numel()
        L.check(L.lib().vmr_cast(g.data_ptr(), L.dtype_code(g), out.data_ptr(), L.dtype_code(out), rows, cols,
                                 lds, cols, 0.0, 0, None, L.stream_ptr()), "vmr_cast")
        return out, None, None


def to_dtype(x, dtype, pad8=False):
    L.require_gpu(x)
    return _ToDtype.apply(x, dtype, pad8)


# ---------------------------------------------------------------------------
# Linear / pointwise Conv1D with fused epilogue
# ---------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    """y = drop(act(x.W^T + b)) + residual   (reference Conv1D, layers.py:15-26).
    `weights` are the fp32 master parameters (a group is concatenated along N);
    the compute-dtype copy comes from the WeightCache."""

    @staticmethod
    def forward(ctx, x, bias, bias2, bias_scale, residual, cache, relu, drop, rowscale, kslice, res_div, tee, res_pre,
                *weights):
        L.require_gpu(x)
        assert x.dim() == 2 and x.stride(1) == 1
        W = cache.get(weights, x.dtype, 0 if kslice is not None else x.shape[1])
        if kslice is not None:               # y = x . W[:, k0:k1]^T : a column slice of ONE weight, used in place
            assert len(weights) == 1 and kslice[0] % 8 == 0 and kslice[1] % 8 == 0
            W = W[:, kslice[0]:kslice[1]]
        M, Kp = x.shape
        N = W.shape[0]
        assert W.shape[1] == Kp, f"input K {Kp} vs weight K {W.shape[1]} (pad inputs to a multiple of 8)"
        dt = L.dtype_code(x)
        Np = _rup(N, 8)                      # keep 16-byte rows even for N = 1 / 4 heads
        ybuf = (torch.zeros if Np != N else torch.empty)(M, Np, device=x.device, dtype=x.dtype)
        y = ybuf[:, :N] if Np != N else ybuf
        flags = 0
        aux = None
        if bias is not None:
            flags |= L.EPI_BIAS
        if relu:
            flags |= L.EPI_RELU
        if drop[0] > 0:
            flags |= L.EPI_DROPOUT
        if residual is not None:
            assert residual.is_contiguous() and residual.shape[1] == N and residual.shape[0] * res_div == M
            flags |= L.EPI_RESIDUAL | (L.EPI_RES_PRE if res_pre else 0)
        res_pre = bool(res_pre and residual is not None)
        aux_bits = False
        if relu and (residual is not None) and not res_pre and any(t.requires_grad for t in (x, *weights)):
            # the backward's ReLU / dropout mask: one BIT per element where the GEMM runs on a register-direct
            # epilogue (the conv-block products: 1.2 MB instead of a 19 MB bf16 copy of the activation at cfg2),
            # otherwise the post-dropout ReLU output itself
            probe = L.GemmDesc()
            probe.A, probe.B, probe.C = x.data_ptr(), W.data_ptr(), ybuf.data_ptr()
            probe.bias, probe.bias2, probe.residual = _ptr(bias), _ptr(bias2), _ptr(residual)
            probe.lda, probe.ldb, probe.ldc, probe.ldr = x.stride(0), W.stride(0), Np, N
            probe.M, probe.N, probe.K, probe.transA, probe.transB = M, N, Kp, 0, 0
            probe.dtype, probe.flags, probe.Z1, probe.Z2, probe.splitk = dt, flags, 1, 1, 1
            aux_bits = AUX_BITS and Np == N and rowscale is None and res_div == 1 and \
                bool(L.lib().vmr_gemm_aux_bits_supported(C.byref(probe)))
            if aux_bits:
                aux = torch.empty(M, N // 8, device=x.device, dtype=torch.uint8)
                flags |= L.EPI_AUX | L.EPI_AUX_BITS
            else:
                aux = torch.empty_like(y)
                flags |= L.EPI_AUX
        if rowscale is not None:
            flags |= L.EPI_ROWSCALE
        few = None
        if (Np == N and not relu and drop[0] == 0 and residual is None and rowscale is None and bias2 is None):
            few = mm_few_tiles(x, W, 0, 0, bias=bias)   # e.g. the pooled-query projection of CQConcatenate: M = B
        if few is not None:
            ybuf = y = few
        else:
            gemm(x, W, ybuf, M, N, Kp, 0, 0, x.stride(0), W.stride(0), Np, dtype=dt, flags=flags, bias=bias,
                 residual=residual, aux=aux, ldr=N, rowscale=rowscale, drop=drop, bias2=bias2, bias_scale=bias_scale,
                 res_div=res_div)
        ctx.save_for_backward(x, W, aux if aux is not None else (ybuf if relu else None), rowscale)
        ctx.meta = (relu, drop, bias is not None, residual is not None, [tuple(w.shape) for w in weights])
        ctx.weights = weights
        ctx.state = cache.state
        ctx.bias_param = bias
        ctx.bias2_param, ctx.bias_scale = bias2, bias_scale
        ctx.kslice, ctx.res_div, ctx.res_pre = kslice, res_div, res_pre
        ctx.aux_bits = aux_bits
        assert res_div == 1 or aux is None
        assert not (res_pre and rowscale is not None)
        assert bias2 is None or (bias is not None and Np == N)
        if tee:   # second output = x itself: its other consumer's gradient arrives here and rides on the dX epilogue
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dxtra=None):
        x, W, h, rowscale = ctx.saved_tensors
        relu, drop, has_bias, has_res, wshapes = ctx.meta
        M, N = dy.shape
        Np = _rup(N, 8)
        if Np != N:                          # narrow heads: work on a zero-padded [M, Np] copy
            assert drop[0] == 0.0, "narrow outputs carry no dropout in this model"
            pad = torch.zeros(M, Np, device=dy.device, dtype=dy.dtype)
            pad[:, :N] = dy
            dyb = pad
        else:
            dyb = dy.contiguous()
        if rowscale is not None:
            dyb = dyb * rowscale[:, None].to(dyb.dtype)
        dres = dyb if has_res else None      # (has_res implies Np == N)
        if has_res and ctx.res_div > 1 and not ctx.res_pre:      # broadcast residual: its gradient is the sum over each row group
            dres = dyb.view(M // ctx.res_div, ctx.res_div, Np).sum(1, dtype=torch.float32).to(dyb.dtype)
        Kp = x.shape[1]
        dt = L.dtype_code(dyb)
        lib, st = L.lib(), L.stream_ptr()
        scale = 1.0 / (1.0 - drop[0]) if drop[0] > 0 else 1.0
        colsum_in_gemm = False
        bgrad = main_grad(ctx.bias_param) if (has_bias and Np == N) else None
        has_b2 = ctx.bias2_param is not None
        if has_b2 and (bgrad is None or main_grad(ctx.bias2_param) is None):
            bgrad = None                      # both bias terms go through autograd or neither
        db = (bgrad if bgrad is not None else torch.zeros(Np, device=dy.device, dtype=torch.float32)) \
            if has_bias else None
        db2 = main_grad(ctx.bias2_param) if (has_b2 and bgrad is not None) else None
        dbs = ctx.bias_scale if bgrad is not None else 1.0   # (autograd path: scaled below)
        if relu:
            dzb = torch.empty_like(dyb)
            L.check(lib.vmr_relu_bwd_bias(3 if ctx.aux_bits else 1, dyb.data_ptr(), h.data_ptr(), dzb.data_ptr(), _ptr(db), M, Np, Np, scale,
                                          dt, 0.0, 0, None, _ptr(db2), dbs, st), "vmr_relu_bwd_bias")
        elif drop[0] > 0:
            dzb = torch.empty_like(dyb)
            L.check(lib.vmr_relu_bwd_bias(2, dyb.data_ptr(), None, dzb.data_ptr(), _ptr(db), M, Np, Np, scale, dt,
                                          drop[0], drop[1], _ptr(drop[2]), _ptr(db2), dbs, st), "vmr_relu_bwd_bias")
        else:
            dzb = dyb
            # plain linear layer: the bias gradient (column sums of dz) rides on the weight-gradient GEMM below
            # (vmr_gemm_t.a_colsum) when that GEMM accumulates straight into the arena
            kf0 = int(np.prod(wshapes[0][1:]))
            colsum_in_gemm = (has_bias and bgrad is not None and db2 is None and
                              all(main_grad(w) is not None for w in ctx.weights) and
                              (ctx.kslice is not None or all(int(np.prod(shp[1:])) == x.shape[1] for shp in wshapes) or
                               (len(wshapes) == 1 and kf0 < x.shape[1] and kf0 % 4 == 0 and USE_SLABS and M >= 256)))
            if has_bias and not colsum_in_gemm:
                L.check(lib.vmr_relu_bwd_bias(0, dyb.data_ptr(), None, None, db.data_ptr(), M, Np, Np, 1.0, dt, 0.0, 0,
                                              None, _ptr(db2), dbs, st), "vmr_relu_bwd_bias")
        dz = dzb[:, :N] if Np != N else dzb
        if has_res and ctx.res_pre:          # the residual sat inside the activation: its gradient is dz, not dy
            dres = dzb
            if ctx.res_div > 1:
                dres = dzb.view(M // ctx.res_div, ctx.res_div, Np).sum(1, dtype=torch.float32).to(dzb.dtype)
        dx = None
        held = None       # the dX product, held back until the first weight-gradient slab product takes it along
        state = ctx.state
        if ctx.needs_input_grad[0]:
            few = mm_few_tiles(dz, W, 0, 1, addend=dxtra)
            if few is not None:
                dx = few
            else:
                # K-major weight copy from the optimizer arena: dX = dz . Wt^T runs on the row-major-weight kernel
                Wt = WeightCache.get_t(ctx.weights, dz.dtype) if Np == N else None
                if Wt is not None and ctx.kslice is not None:
                    Wt = Wt[ctx.kslice[0]:ctx.kslice[1]]
                if Wt is not None and tuple(Wt.shape) != (Kp, N):
                    Wt = None                # (K-padded inputs use a padded cast copy of W, not the arena mirror)
                Bm, tb = (Wt, 0) if Wt is not None else (W, 1)
                # (held back: launched together with the first weight-gradient slab product below, vmr_gemm2)
                hold = Wt is not None and DW_SIDE_STREAM is None and MERGE_DX_DW
                if dxtra is not None:        # tee: dX = dz.W + (gradient of x's other consumer), one epilogue
                    dxtra = dxtra.contiguous()
                    assert dxtra.shape == x.shape and dxtra.dtype == dz.dtype
                    dx = mm(dz, Bm, 0, tb, flags=L.EPI_RESIDUAL, residual=dxtra, ldr=dxtra.stride(0), hold=hold)
                else:
                    dx = mm(dz, Bm, 0, tb, hold=hold)                                          # [M,N] . [N,Kp]
                if hold:
                    dx, held = dx
        elif dxtra is not None:
            dx = dxtra
        # dW = dz^T . x  -> fp32 [N,Kp], split-K over the M (token) dimension
        sk = splitk_for(N, Kp, M)
        slots = [main_grad(w) for w in ctx.weights]
        ks = ctx.kslice
        kfull = [int(np.prod(shp[1:])) for shp in wshapes]
        # (a single weight whose K was zero-padded for the GEMM -- V = 500 -> 512, 400 -> 512 -- takes the arena path too:
        #  its slab rows are Kp wide, the reduction writes the first K columns of each into the dense [N, K] slot)
        kpadded = ks is None and len(slots) == 1 and kfull[0] < Kp and kfull[0] % 4 == 0 and USE_SLABS and M >= 256
        if all(g is not None for g in slots) and (ks is not None or all(k == Kp for k in kfull) or kpadded):
            # accumulate straight into the flat gradient arena: no zero-fill, no autograd add
            side = DW_SIDE_STREAM
            if side is not None:
                side.wait_stream(torch.cuda.current_stream())
                dz.record_stream(side)
                x.record_stream(side)
            # a grouped projection (q|k|v, ...) whose gradient slots sit back to back in the arena gets ONE
            # [sum N, K] weight-gradient product (and one slab reduction) instead of one per member
            if GROUP_DW and len(slots) > 1 and ks is None and all(k == Kp for k in kfull) and \
                    all(a_.data_ptr() + a_.numel() * 4 == b_.data_ptr() for a_, b_ in zip(slots[:-1], slots[1:])):
                ntot = sum(shp[0] for shp in wshapes)
                slots = [torch.as_strided(slots[0], (ntot, Kp), (Kp, 1))]
                wshapes_, kfull_ = [(ntot, Kp)], [Kp]
            else:
                wshapes_, kfull_ = wshapes, kfull
            with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                r = 0
                for g_, shp, kf in zip(slots, wshapes_, kfull_):
                    n = shp[0]
                    a = dz[:, r:r + n]
                    # (a held-back single-round dX product will share this launch: fewer, longer splits)
                    merged = held is not None and _cdiv(M, 160) * _cdiv(Kp, 128) <= 512
                    sk_ = splitk_for(n, Kp, M, SPLITK_TARGET_MERGED if merged else 0)
                    if kpadded:
                        sk_ = max(2, sk_)    # (always through slabs: a direct accumulation would need ldc = K < Kp)
                    if ks is not None:       # gradient of the column slice, in place inside the full matrix
                        g_ = g_.view(n, kf)[:, ks[0]:ks[1]]
                    if sk_ > 1 and USE_SLABS:
                        # split-K partials as plain fp32 slabs + one reduce pass: float atomics from
                        # every workgroup of a single-round grid land together and run far below HBM speed
                        ws = torch.empty(sk_, n, Kp, device=dy.device, dtype=torch.float32)
                        gemm(a, x, ws, n, Kp, M, 1, 1, a.stride(0), x.stride(0), Kp, dtype=dt, flags=L.EPI_SLAB,
                             splitk=sk_, a_colsum=db[r:r + n] if colsum_in_gemm else None, held=held, state=state)
                        state.reduce_later(ws, g_, sk_, n * Kp, Kp, kf)
                    else:
                        gemm(a, x, g_, n, Kp, M, 1, 1, a.stride(0), x.stride(0), kf, dtype=dt, flags=L.EPI_ACCUM,
                             splitk=sk_, a_colsum=db[r:r + n] if colsum_in_gemm else None, held=held)
                    held = None
                    r += n
            if db is not None and Np != N:
                db = db[:N]
            return (dx, *_bias_grads(db, bgrad, has_b2, ctx.bias_scale), None, dres, None, None, None, None, None, None,
                    None, None, *([None] * len(wshapes)))
        if held is not None:
            held.launch()
        if sk > 1:
            dW = torch.zeros(N, Kp, device=dy.device, dtype=torch.float32)
            gemm(dz, x, dW, N, Kp, M, 1, 1, dz.stride(0), x.stride(0), Kp, dtype=dt, flags=L.EPI_ACCUM, splitk=sk)
        else:
            dW = mm(dz, x, 1, 1, out_f32=True)
        grads = []
        r = 0
        for shp in wshapes:
            n = shp[0]
            k = 1
            for s_ in shp[1:]:
                k *= s_
            if ks is not None:
                full = torch.zeros(n, k, device=dy.device, dtype=torch.float32)
                full[:, ks[0]:ks[1]] = dW[r:r + n, :ks[1] - ks[0]]
                grads.append(full.reshape(shp))
            else:
                grads.append(dW[r:r + n, :k].reshape(shp))
            r += n
        if db is not None and Np != N:
            db = db[:N]
        return (dx, *_bias_grads(db, bgrad, has_b2, ctx.bias_scale), None, dres, None, None, None, None, None, None,
                None, None, *grads)


def _bias_grads(db, bgrad, has_b2, bias_scale):
    """(d bias, d bias2) handed back to autograd: nothing when the kernel accumulated into the arena."""
    if db is None or bgrad is not None:
        return None, None
    return (db * bias_scale if has_b2 else db), (db if has_b2 else None)


def linear(x, weights, bias, cache, *, relu=False, drop=NO_DROP, residual=None, rowscale=None, bias2=None,
           bias_scale=1.0, kslice=None, res_div=1, tee=False, res_pre=False):
    """y = drop(act(x.W[:, kslice]^T + bias_scale*bias + bias2)) + residual[row // res_div].
    res_pre=True moves the residual inside the activation: y = drop(act(x.W^T + bias + residual)).
    tee=True returns (y, x): hand that x to the tensor's OTHER consumer, whose gradient then joins dX inside the
    backward GEMM's epilogue instead of through a separate autograd add pass."""
    if isinstance(weights, torch.Tensor):
        weights = [weights]
    return _Linear.apply(x, bias, bias2, bias_scale, residual, cache, relu, drop, rowscale, kslice, res_div, tee,
                         res_pre, *weights)


class _WeightedPool(torch.autograd.Function):
    """WeightedPool (reference models/layers.py:440-453) in one kernel each way."""

    @staticmethod
    def forward(ctx, x, w, mask):
        L.require_gpu(x, w, mask)
        B, Ls, D = x.shape
        x = x.contiguous()
        alpha = torch.empty(B, Ls, device=x.device, dtype=torch.float32)
        pooled = torch.empty(B, D, device=x.device, dtype=x.dtype)
        L.check(L.lib().vmr_weighted_pool_fwd(x.data_ptr(), w.data_ptr(), mask.data_ptr(), alpha.data_ptr(),
                                              pooled.data_ptr(), B, Ls, D, L.dtype_code(x), L.stream_ptr()),
                "vmr_weighted_pool_fwd")
        ctx.save_for_backward(x, w, alpha)
        return pooled

    @staticmethod
    def backward(ctx, dp):
        x, w, alpha = ctx.saved_tensors
        B, Ls, D = x.shape
        dp = dp.contiguous()
        dx = torch.empty_like(x)
        mg = main_grad(w)
        dw = mg if mg is not None else torch.zeros(w.numel(), device=x.device, dtype=torch.float32)
        L.check(L.lib().vmr_weighted_pool_bwd(dp.data_ptr(), x.data_ptr(), w.data_ptr(), alpha.data_ptr(), dx.data_ptr(),
                                              dw.data_ptr(), B, Ls, D, L.dtype_code(x), L.stream_ptr()),
                "vmr_weighted_pool_bwd")
        return dx, (None if mg is not None else dw.view(w.shape)), None


def weighted_pool(x, w, mask):
    """x [B,L,D] (compute dtype), w [D,1] fp32 parameter, mask [B,L] fp32 -> pooled [B,D]."""
    return _WeightedPool.apply(x, w, mask.contiguous())


class _NarrowLinear(torch.autograd.Function):
    """Conv1D with <= 8 output channels (the match / start / end heads) on the matrix-vector kernels:
    fp32 logits [M, N] straight from the compute-dtype features, one backward pass for dx, dW, db."""

    @staticmethod
    def forward(ctx, x, W, bias, N, tee=False):
        L.require_gpu(x, W)
        assert x.dim() == 2 and x.stride(1) == 1
        M, K = x.shape
        Wm = W.detach().reshape(N, -1)
        assert Wm.shape[1] == K and Wm.is_contiguous()
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
        L.check(L.lib().vmr_narrow_linear_fwd(x.data_ptr(), Wm.data_ptr(), _ptr(bias), y.data_ptr(), M, N, K, x.stride(0),
                                              L.dtype_code(x), L.stream_ptr()), "vmr_narrow_linear_fwd")
        ctx.save_for_backward(x, W, bias)
        ctx.N = N
        if tee:   # second output = x itself: its other consumer's gradient then joins dx inside the dx kernel
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dxtra=None):
        x, W, bias = ctx.saved_tensors
        M, K = x.shape
        N = ctx.N
        dy = dy.contiguous().float()
        dx = torch.empty(M, K, device=x.device, dtype=x.dtype) if ctx.needs_input_grad[0] else None
        if dxtra is not None:
            dxtra = dxtra.contiguous()
            assert dx is not None and dxtra.shape == x.shape and dxtra.dtype == x.dtype and x.is_contiguous()
        gW, gb = main_grad(W), (main_grad(bias) if bias is not None else None)
        dW = gW if gW is not None else torch.zeros(N, K, device=x.device, dtype=torch.float32)
        db = None
        if bias is not None:
            db = gb if gb is not None else torch.zeros(N, device=x.device, dtype=torch.float32)
        groups = max(1, 256 // (K // 8))
        ws = torch.empty(_cdiv(M, 32) * groups * (N * K + N), device=x.device, dtype=torch.float32)   # VMR_NARROW_WS_FLOATS
        L.check(L.lib().vmr_narrow_linear_bwd_add(dy.data_ptr(), x.data_ptr(), W.detach().reshape(N, -1).data_ptr(), _ptr(dx),
                                                  _ptr(dxtra), dW.data_ptr(), _ptr(db), ws.data_ptr(), M, N, K, x.stride(0),
                                                  L.dtype_code(x), L.stream_ptr()), "vmr_narrow_linear_bwd_add")
        return (dx, None if gW is not None else dW.reshape(W.shape),
                None if (bias is None or gb is not None) else db, None, None)


class _LabelFuse(torch.autograd.Function):
    """fuse2 = (fuse + match_score . label_embs^T) * vmask (reference models/SeqPAN.py:80-82) as a rank-4 update of the
    streamed [tokens, D] matrix instead of K = 8 / N = 8 products on 128-wide MFMA tiles.  probs fp32 [M,4], E fp32 [D,4]."""

    @staticmethod
    def forward(ctx, res, probs, E, rowscale):
        L.require_gpu(res, probs, E)
        res, probs = res.contiguous(), probs.contiguous().float()
        M, K = res.shape
        N = probs.shape[1]
        assert tuple(E.shape) == (K, N) and E.is_contiguous() and E.dtype == torch.float32
        y = torch.empty_like(res)
        L.check(L.lib().vmr_label_fuse_fwd(probs.data_ptr(), E.data_ptr(), res.data_ptr(), _ptr(rowscale), y.data_ptr(), M, N, K,
                                           L.dtype_code(res), L.stream_ptr()), "vmr_label_fuse_fwd")
        ctx.save_for_backward(probs, E, rowscale)
        ctx.E_param = E
        return y

    @staticmethod
    def backward(ctx, dy):
        probs, E, rowscale = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = dy.shape
        N = probs.shape[1]
        dres = torch.empty_like(dy)
        dp = torch.empty(M, N, device=dy.device, dtype=torch.float32)
        gE = main_grad(ctx.E_param)
        direct = gE is not None and gE.is_contiguous() and tuple(gE.shape) == (K, N)
        dE = gE if direct else torch.zeros(K, N, device=dy.device, dtype=torch.float32)
        groups = max(1, 256 // (K // 8))
        ws = torch.empty(_cdiv(M, 32) * groups * (N * K + N), device=dy.device, dtype=torch.float32)   # VMR_NARROW_WS_FLOATS
        L.check(L.lib().vmr_label_fuse_bwd(dy.data_ptr(), probs.data_ptr(), E.data_ptr(), _ptr(rowscale), dres.data_ptr(),
                                           dp.data_ptr(), dE.data_ptr(), ws.data_ptr(), M, N, K, L.dtype_code(dy), L.stream_ptr()),
                "vmr_label_fuse_bwd")
        return dres, dp, (None if direct else dE), None


def label_fuse(res, probs, E, rowscale=None):
    return _LabelFuse.apply(res, probs, E, rowscale)


def narrow_linear(x, W, bias, N=None, tee=False):
    """x [M,K] (compute dtype) . W^T + bias -> fp32 [M,N], N <= 8; W holds N*K values as N contiguous rows
    ([N,K,1] conv weights; a [K,1] column vector with N=1).  tee=True: returns (y, alias of x) -- see _Linear."""
    return _NarrowLinear.apply(x, W, bias, W.shape[0] if N is None else N, tee)


class _GumbelSoftmax(torch.autograd.Function):
    """F.gumbel_softmax(logits, tau) (reference models/SeqPAN.py:79) over <= 8 classes in one kernel, plus the
    zero-padded compute-dtype copy of the probabilities that feeds the label-embedding product."""

    @staticmethod
    def forward(ctx, logits, noise, tau, seed, step, pad_to, dtype):
        L.require_gpu(logits)
        R, Cc = logits.shape
        logits = logits.contiguous().float()
        probs = torch.empty(R, Cc, device=logits.device, dtype=torch.float32)
        padded = torch.empty(R, pad_to, device=logits.device, dtype=dtype)
        L.check(L.lib().vmr_gumbel_softmax_fwd(logits.data_ptr(), _ptr(noise), tau, seed, _ptr(step), probs.data_ptr(),
                                               padded.data_ptr(), R, Cc, pad_to, L.dtype_code(padded), L.stream_ptr()),
                "vmr_gumbel_softmax_fwd")
        ctx.save_for_backward(probs)
        ctx.meta = (tau, pad_to, dtype)
        ctx.set_materialize_grads(False)     # an unused output must not cost a zero-filled gradient
        return probs, padded

    @staticmethod
    def backward(ctx, dprobs, dpadded):
        (probs,) = ctx.saved_tensors
        tau, pad_to, dtype = ctx.meta
        R, Cc = probs.shape
        if dprobs is None and dpadded is None:
            return None, None, None, None, None, None, None
        dprobs = None if dprobs is None else dprobs.contiguous().float()
        dpadded = None if dpadded is None else dpadded.contiguous()
        dl = torch.empty_like(probs)
        L.check(L.lib().vmr_gumbel_softmax_bwd(_ptr(dprobs), _ptr(dpadded), probs.data_ptr(), tau, dl.data_ptr(), R, Cc, pad_to,
                                               L.dtype_code(dpadded) if dpadded is not None else L.BF16, L.stream_ptr()),
                "vmr_gumbel_softmax_bwd")
        return dl, None, None, None, None, None, None


def gumbel_softmax(logits, noise, tau, seed, step, pad_to, dtype):
    """logits fp32 [R,C<=8]; noise: explicit Gumbel noise [R,C] or None (drawn in the kernel from seed/step).
    Returns (probs fp32 [R,C], padded `dtype` [R,pad_to])."""
    if noise is not None:
        noise = noise.contiguous().float()
    return _GumbelSoftmax.apply(logits, noise, float(tau), int(seed), step, int(pad_to), dtype)


class _MatchLoss(torch.autograd.Function):
    """lossfun_match (reference models/loss.py:24-41), forward and backward in one launch each."""

    @staticmethod
    def forward(ctx, probs, label_embs, labels, vmask):
        L.require_gpu(probs, label_embs, labels, vmask)
        Cc = probs.shape[-1]
        p2 = probs.contiguous().float().view(-1, Cc)
        R = p2.shape[0]
        E = label_embs.detach().contiguous().float()
        D = E.shape[0]
        assert E.shape[1] == Cc
        lab = labels.contiguous().view(-1).long()
        vm = vmask.contiguous().float().view(-1)
        assert lab.numel() == R and vm.numel() == R
        loss = torch.empty(1, device=p2.device, dtype=torch.float32)
        aux = torch.empty(Cc * Cc + 2 + L.MATCH_LOSS_SCRATCH, device=p2.device, dtype=torch.float32)
        L.check(L.lib().vmr_match_loss_fwd(p2.data_ptr(), lab.data_ptr(), vm.data_ptr(), E.data_ptr(), loss.data_ptr(),
                                           aux.data_ptr(), R, D, Cc, L.stream_ptr()), "vmr_match_loss_fwd")
        ctx.save_for_backward(lab, vm, label_embs, aux)
        ctx.pshape = probs.shape
        return loss.view(())

    @staticmethod
    def backward(ctx, dloss):
        lab, vm, label_embs, aux = ctx.saved_tensors
        Cc = ctx.pshape[-1]
        R = lab.numel()
        E = label_embs.detach().contiguous().float()
        D = E.shape[0]
        dl = dloss.contiguous().float().view(1)
        dprobs = torch.empty(R, Cc, device=dl.device, dtype=torch.float32)
        gE = main_grad(label_embs)
        need_e = ctx.needs_input_grad[1]
        dE = gE if (gE is not None and gE.is_contiguous()) else (torch.zeros(D, Cc, device=dl.device, dtype=torch.float32)
                                                              if need_e else None)
        L.check(L.lib().vmr_match_loss_bwd(dl.data_ptr(), lab.data_ptr(), vm.data_ptr(), E.data_ptr(), aux.data_ptr(),
                                           dprobs.data_ptr(), _ptr(dE), R, D, Cc, L.stream_ptr()), "vmr_match_loss_bwd")
        return dprobs.view(ctx.pshape), (None if (dE is None or dE is gE) else dE), None, None


def match_loss(probs, label_embs, labels, vmask):
    return _MatchLoss.apply(probs, label_embs, labels, vmask)


class _ScaleShift(torch.autograd.Function):
    """y = x * a + b over the last dim (a, b fp32 parameters): the rank-1-folded operand of the CQAttention
    trilinear score (reference models/layers.py:427-437), one kernel each way."""

    @staticmethod
    def forward(ctx, x, a, b):
        L.require_gpu(x, a, b)
        D = x.shape[-1]
        x = x.contiguous()
        y = torch.empty_like(x)
        L.check(L.lib().vmr_scale_shift_fwd(x.data_ptr(), a.data_ptr(), b.data_ptr(), y.data_ptr(), x.numel() // D, D,
                                            L.dtype_code(x), L.stream_ptr()), "vmr_scale_shift_fwd")
        ctx.save_for_backward(x, a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, a, b = ctx.saved_tensors
        D = x.shape[-1]
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        ga, gb = main_grad(a), main_grad(b)
        da = ga if ga is not None else torch.zeros(D, device=x.device, dtype=torch.float32)
        db = gb if gb is not None else torch.zeros(D, device=x.device, dtype=torch.float32)
        L.check(L.lib().vmr_scale_shift_bwd(dy.data_ptr(), x.data_ptr(), a.data_ptr(), dx.data_ptr(), da.data_ptr(),
                                            db.data_ptr(), x.numel() // D, D, L.dtype_code(x), L.stream_ptr()),
                "vmr_scale_shift_bwd")
        return dx, (None if ga is not None else da.view(a.shape)), (None if gb is not None else db.view(b.shape))


def scale_shift(x, a, b):
    """x [..., D] (compute dtype) * a + b; a, b: fp32 parameters with D elements (any shape)."""
    assert a.numel() == x.shape[-1] and b.numel() == x.shape[-1] and a.is_contiguous() and b.is_contiguous()
    return _ScaleShift.apply(x, a, b)


class _CharCnn(torch.autograd.Function):
    """CharacterEmbedding (reference models/layers.py:51-75): lookup + dropout + 4 x (conv (1,k) + ReLU + max over
    positions) in one kernel each way (csrc/charcnn.hip)."""

    @staticmethod
    def forward(ctx, char_ids, table, drop, dtype, *wb):
        L.require_gpu(char_ids, table)
        ws_, bs_ = wb[:4], wb[4:]
        ids = char_ids.contiguous()
        Wn = ids.numel() // ids.shape[-1]
        Cc = ids.shape[-1]
        CD = table.shape[1]
        oc = [w.shape[0] for w in ws_]
        for k, w in enumerate(ws_):
            assert tuple(w.shape) == (oc[k], CD, 1, k + 1) and w.is_contiguous()
        OT = sum(oc)
        out = torch.empty(Wn, OT, device=table.device, dtype=dtype)
        amax = torch.empty(Wn, OT, device=table.device, dtype=torch.int8)
        wp = (C.c_void_p * 4)(*[w.data_ptr() for w in ws_])
        bp = (C.c_void_p * 4)(*[b.data_ptr() for b in bs_])
        ocp = (C.c_int * 4)(*oc)
        L.check(L.lib().vmr_char_cnn_fwd(ids.data_ptr(), table.data_ptr(), wp, bp, ocp, out.data_ptr(), OT, amax.data_ptr(), Wn,
                                         Cc, CD, L.dtype_code(out), drop[0], drop[1], _ptr(drop[2]), L.stream_ptr()),
                "vmr_char_cnn_fwd")
        ctx.save_for_backward(ids, table, out, amax, *wb)
        ctx.meta = (drop, oc, Wn, Cc, CD, OT)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, table, out, amax, *wb = ctx.saved_tensors
        drop, oc, Wn, Cc, CD, OT = ctx.meta
        ws_, bs_ = wb[:4], wb[4:]
        dout = dout.contiguous()
        dev = out.device
        gts = [main_grad(t) for t in (*ws_, *bs_)]
        direct = all(g is not None for g in gts)
        gws = [g if direct else torch.zeros_like(t, dtype=torch.float32) for g, t in zip(gts[:4], ws_)]
        gbs = [g if direct else torch.zeros_like(t, dtype=torch.float32) for g, t in zip(gts[4:], bs_)]
        gtab = main_grad(table)
        dtab = gtab if gtab is not None else torch.zeros_like(table, dtype=torch.float32)
        ocp = (C.c_int * 4)(*oc)
        nws = L.lib().vmr_char_cnn_ws_floats(Wn, CD, ocp, L.dtype_code(out))
        ws = torch.empty(nws, device=dev, dtype=torch.float32)
        wp = (C.c_void_p * 4)(*[w.data_ptr() for w in ws_])
        bp = (C.c_void_p * 4)(*[b.data_ptr() for b in bs_])
        dwp = (C.c_void_p * 4)(*[g.data_ptr() for g in gws])
        dbp = (C.c_void_p * 4)(*[g.data_ptr() for g in gbs])
        L.check(L.lib().vmr_char_cnn_bwd(dout.data_ptr(), out.data_ptr(), OT, amax.data_ptr(), ids.data_ptr(), table.data_ptr(),
                                         wp, bp, ocp, dwp, dbp, dtab.data_ptr(), ws.data_ptr(), Wn, Cc, CD, L.dtype_code(out),
                                         drop[0], drop[1], _ptr(drop[2]), L.stream_ptr()), "vmr_char_cnn_bwd")
        gw_ret = [None] * 8 if direct else [*gws, *gbs]
        return (None, None if gtab is not None else dtab, None, None, *gw_ret)


class _TextEmbed(torch.autograd.Function):
    """Embedding.forward up to the concat (reference models/layers.py:87-91): word lookup (+ dropout) and the character
    CNN write their column ranges of ONE [words, ldo] matrix in the compute dtype -- [0, wd) words, [wd, wd + OT) chars,
    the rest zero (K padding of query_conv1d) -- two launches, no table concat / cast / cat / zero-fill glue.  Only
    unk_vec, the character table and the character convolutions receive gradients (pad_vec / glove_vec are frozen)."""

    @staticmethod
    def forward(ctx, word_ids, char_ids, pad_vec, unk_vec, glove, table, drop_w, drop_c, dtype, ldo, *wb):
        L.require_gpu(word_ids, char_ids, glove, table)
        ws_, bs_ = wb[:4], wb[4:]
        wid, cid = word_ids.contiguous(), char_ids.contiguous()
        n, wd = wid.numel(), glove.shape[1]
        Cc, CD = cid.shape[-1], table.shape[1]
        oc = [w.shape[0] for w in ws_]
        OT = sum(oc)
        assert cid.numel() // Cc == n and ldo >= wd + OT and wd % 4 == 0
        out = torch.empty(n, ldo, device=glove.device, dtype=dtype)
        dt = L.dtype_code(out)
        lib, st = L.lib(), L.stream_ptr()
        L.check(lib.vmr_word_embedding_fwd(wid.data_ptr(), pad_vec.data_ptr(), unk_vec.data_ptr(), glove.data_ptr(), out.data_ptr(),
                                           n, wd, glove.shape[0], ldo, wd + OT, ldo, dt, drop_w[0], drop_w[1], _ptr(drop_w[2]), st),
                "vmr_word_embedding_fwd")
        amax = torch.empty(n, OT, device=table.device, dtype=torch.int8)
        wp = (C.c_void_p * 4)(*[w.data_ptr() for w in ws_])
        bp = (C.c_void_p * 4)(*[b.data_ptr() for b in bs_])
        ocp = (C.c_int * 4)(*oc)
        cout = out[:, wd:]                                   # the character columns: same row stride
        L.check(lib.vmr_char_cnn_fwd(cid.data_ptr(), table.data_ptr(), wp, bp, ocp, cout.data_ptr(), ldo, amax.data_ptr(), n, Cc, CD,
                                     dt, drop_c[0], drop_c[1], _ptr(drop_c[2]), st), "vmr_char_cnn_fwd")
        ctx.save_for_backward(wid, cid, table, out, amax, unk_vec, *wb)
        ctx.meta = (drop_w, drop_c, oc, n, wd, Cc, CD, OT, ldo)
        return out

    @staticmethod
    def backward(ctx, dout):
        wid, cid, table, out, amax, unk_vec, *wb = ctx.saved_tensors
        drop_w, drop_c, oc, n, wd, Cc, CD, OT, ldo = ctx.meta
        ws_, bs_ = wb[:4], wb[4:]
        dout = dout.contiguous()
        dev, dt = out.device, L.dtype_code(out)
        lib, st = L.lib(), L.stream_ptr()
        gu = main_grad(unk_vec)
        dunk = gu if gu is not None else torch.zeros_like(unk_vec, dtype=torch.float32)
        L.check(lib.vmr_word_embedding_bwd(wid.data_ptr(), dout.data_ptr(), dunk.data_ptr(), n, wd, ldo, dt, drop_w[0], drop_w[1],
                                           _ptr(drop_w[2]), st), "vmr_word_embedding_bwd")
        gts = [main_grad(t) for t in (*ws_, *bs_)]
        direct = all(g is not None for g in gts)
        gws = [g if direct else torch.zeros_like(t, dtype=torch.float32) for g, t in zip(gts[:4], ws_)]
        gbs = [g if direct else torch.zeros_like(t, dtype=torch.float32) for g, t in zip(gts[4:], bs_)]
        gtab = main_grad(table)
        dtab = gtab if gtab is not None else torch.zeros_like(table, dtype=torch.float32)
        ocp = (C.c_int * 4)(*oc)
        ws = torch.empty(lib.vmr_char_cnn_ws_floats(n, CD, ocp, dt), device=dev, dtype=torch.float32)
        wp = (C.c_void_p * 4)(*[w.data_ptr() for w in ws_])
        bp = (C.c_void_p * 4)(*[b.data_ptr() for b in bs_])
        dwp = (C.c_void_p * 4)(*[g.data_ptr() for g in gws])
        dbp = (C.c_void_p * 4)(*[g.data_ptr() for g in gbs])
        L.check(lib.vmr_char_cnn_bwd(dout[:, wd:].data_ptr(), out[:, wd:].data_ptr(), ldo, amax.data_ptr(), cid.data_ptr(),
                                     table.data_ptr(), wp, bp, ocp, dwp, dbp, dtab.data_ptr(), ws.data_ptr(), n, Cc, CD, dt,
                                     drop_c[0], drop_c[1], _ptr(drop_c[2]), st), "vmr_char_cnn_bwd")
        gw_ret = [None] * 8 if direct else [*gws, *gbs]
        return (None, None, None, None if gu is not None else dunk, None, None if gtab is not None else dtab, None, None, None,
                None, *gw_ret)


def text_embed(word_ids, char_ids, pad_vec, unk_vec, glove, table, conv_weights, conv_biases, drop_w, drop_c, dtype, ldo):
    """[words, ldo] = [dropout(word vectors) | char-CNN features | 0] in `dtype` (the input of query_conv1d)."""
    return _TextEmbed.apply(word_ids, char_ids, pad_vec, unk_vec, glove, table, drop_w, drop_c, dtype, ldo,
                            *conv_weights, *conv_biases)


def char_cnn(char_ids, table, conv_weights, conv_biases, drop, dtype):
    """char_ids int64 [..., C]; table fp32 [num_chars, char_dim]; 4 conv weights [10k, char_dim, 1, k] + biases
    -> [words, sum(out channels)] in `dtype`."""
    return _CharCnn.apply(char_ids, table, drop, dtype, *conv_weights, *conv_biases)


def group_view(params):
    """ONE flat fp32 view over parameters laid out back to back in the flat arena (optim.FlatArena
    places the members of model.weight_groups() consecutively), carrying the matching view of the
    gradient arena so bias-gradient kernels accumulate in place.  None if the parameters are not
    contiguous (no arena yet): callers fall back to torch.cat."""
    grads = [main_grad(p) for p in params]
    if any(g is None for g in grads):
        return None
    for seq in (params, grads):
        for a_, b_ in zip(seq[:-1], seq[1:]):
            if a_.data_ptr() + a_.numel() * 4 != b_.data_ptr():
                return None
    n = sum(p.numel() for p in params)
    v = torch.as_strided(params[0].detach(), (n,), (1,))
    v._vmr_main_grad = torch.as_strided(grads[0], (n,), (1,))
    return v


# ---------------------------------------------------------------------------
# LayerNorm (+pos, +dropout)
# ---------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    """tee=True also returns an alias of x: use it for the residual branch, so the gradient of
    that branch arrives here and is added inside the LN-backward kernel (dres) instead of by a
    separate elementwise pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, pos, S, drop, cache, tee=False, out=None):
        L.require_gpu(x)
        x = x.contiguous()
        rows, D = x.shape
        if out is not None:      # (rows of a caller-owned packed matrix: see pack_rows)
            assert tuple(out.shape) == tuple(x.shape) and out.dtype == x.dtype and out.is_contiguous() and not tee
        y = out if out is not None else torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        posc = None
        if pos is not None:
            posc = cache.get([pos], x.dtype)
            assert posc.shape[1] == D and S <= posc.shape[0]
        L.check(L.lib().vmr_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, _ptr(posc), S,
                                          y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), rows, D, L.dtype_code(x),
                                          drop[0], drop[1], _ptr(drop[2]), L.stream_ptr()), "vmr_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.meta = (drop, S, None if pos is None else tuple(pos.shape))
        ctx.params = (gamma, beta)
        ctx.pos_param = pos
        ctx.state = cache.state
        if tee:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, mean, rstd = ctx.saved_tensors
        drop, S, pshape = ctx.meta
        dy = dy.contiguous()
        dres = None if dres is None else dres.contiguous()
        rows, D = x.shape
        dx = torch.empty_like(x)
        mg, mb = main_grad(ctx.params[0]), main_grad(ctx.params[1])
        direct = mg is not None and mb is not None
        dg = mg if direct else torch.zeros(D, device=x.device, dtype=torch.float32)
        db = mb if direct else torch.zeros_like(dg)
        # (the positional-table gradient accumulates in place in the arena when there is one: no zero-fill, no add pass)
        mp = main_grad(ctx.pos_param) if pshape else None
        pos_direct = mp is not None and mp.is_contiguous() and tuple(mp.shape) == tuple(pshape)
        dpos = (mp if pos_direct else torch.zeros(pshape, device=x.device, dtype=torch.float32)) if pshape else None
        dpos_ret = None if pos_direct else dpos
        ws = torch.empty(L.ln_bwd_ws_floats(rows, D), device=x.device, dtype=torch.float32)
        if direct and DEFER_COLREDUCE:
            # gradients accumulate into the arena: leave the partial rows in ws, reduce them with every other
            # deferred reduction of this backward pass in one launch
            nb = C.c_int32(0)
            L.check(L.lib().vmr_layernorm_bwd_deferred(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                                       rstd.data_ptr(), _ptr(dres), dx.data_ptr(), _ptr(dpos), ws.data_ptr(),
                                                       S, rows, D, L.dtype_code(x), drop[0], drop[1], _ptr(drop[2]),
                                                       C.byref(nb), L.stream_ptr()), "vmr_layernorm_bwd_deferred")
            ctx.state.defer_colreduce(ws, dg, db, nb.value, D, D, 512 if D <= 512 else (1024 if D <= 1024 else 2048))
            return dx, None, None, None, dpos_ret, None, None, None, None, None
        L.check(L.lib().vmr_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), _ptr(dres), dx.data_ptr(), dg.data_ptr(), db.data_ptr(),
                                          _ptr(dpos), ws.data_ptr(), S, rows, D, L.dtype_code(x), drop[0], drop[1],
                                          _ptr(drop[2]),
                                          L.stream_ptr()), "vmr_layernorm_bwd")
        return dx, (None if direct else dg), (None if direct else db), None, dpos_ret, None, None, None, None, None


class _PackRows(torch.autograd.Function):
    """X = [a; b] where a and b were WRITTEN INTO the two row ranges of X by their producers (layer_norm(out=...)):
    the packed token matrix without a concat pass; backward hands the two row ranges of dX back as views."""

    @staticmethod
    def forward(ctx, X, a, b):
        n0 = a.shape[0]
        assert X.shape[0] == n0 + b.shape[0] and a.data_ptr() == X.data_ptr() and \
            b.data_ptr() == X.data_ptr() + n0 * X.stride(0) * X.element_size()
        ctx.n0 = n0
        return X.view_as(X)

    @staticmethod
    def backward(ctx, dX):
        return None, dX[:ctx.n0], dX[ctx.n0:]


def pack_rows(X, a, b):
    return _PackRows.apply(X, a, b)


class _AddPos(torch.autograd.Function):
    """y = x + pos[row % S] (FeatureEncoderPredict's positional add, reference layers.py:626-631) on the fp32 table
    itself; backward: dx = dy unchanged, the table gradient accumulates in place in the arena when there is one."""

    @staticmethod
    def forward(ctx, x, pos, S):
        L.require_gpu(x, pos)
        x = x.contiguous()
        rows, D = x.shape
        assert pos.dim() == 2 and pos.shape[1] == D and pos.shape[0] >= S and pos.is_contiguous() and pos.dtype == torch.float32
        y = torch.empty_like(x)
        L.check(L.lib().vmr_add_pos_fwd(x.data_ptr(), pos.data_ptr(), y.data_ptr(), rows, S, D, L.dtype_code(x), L.stream_ptr()),
                "vmr_add_pos_fwd")
        ctx.pos_param, ctx.S = pos, S
        return y

    @staticmethod
    def backward(ctx, dy):
        pos, S = ctx.pos_param, ctx.S
        dy = dy.contiguous()
        rows, D = dy.shape
        mp = main_grad(pos)
        direct = mp is not None and mp.is_contiguous() and tuple(mp.shape) == tuple(pos.shape)
        dpos = mp if direct else torch.zeros(pos.shape, device=dy.device, dtype=torch.float32)
        L.check(L.lib().vmr_add_pos_bwd(dy.data_ptr(), dpos.data_ptr(), rows, S, D, L.dtype_code(dy), L.stream_ptr()),
                "vmr_add_pos_bwd")
        return dy, (None if direct else dpos), None


def add_pos(x, pos, S):
    return _AddPos.apply(x, pos, S)


def layer_norm(x, gamma, beta, eps, cache, *, pos=None, S=0, drop=NO_DROP, tee=False, out=None):
    return _LayerNorm.apply(x, gamma, beta, eps, pos, S, drop, cache, tee, out)


# ---------------------------------------------------------------------------
# fused LayerNorm + depthwise conv (k=7)
# ---------------------------------------------------------------------------
class _LnDwConv(torch.autograd.Function):
    """u = dwconv7(LN(x)) per sequence (reference layers.py:139-145, unmasked).
    x is a packed token matrix [sum_i B_i*S_i, D]; `segs` lists the (B_i, S_i)
    sequence groups stored back to back (video clips, then query sentences)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w, eps, segs, tee=False, cache=None):
        L.require_gpu(x)
        x = x.contiguous()
        rows, D = x.shape
        assert rows == sum(b * s for b, s in segs)
        ctx.state = None if cache is None else cache.state
        u = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        w2 = w.detach().reshape(D, 7).contiguous()
        r = 0
        for i in range(0, len(segs), 2):   # two sequence groups (clips + sentences) per launch
            (B1, S1), (B2, S2) = segs[i], (segs[i + 1] if i + 1 < len(segs) else (0, 0))
            L.check(L.lib().vmr_ln_dwconv_fwd2(x[r:].data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, w2.data_ptr(),
                                               u[r:].data_ptr(), mean[r:].data_ptr(), rstd[r:].data_ptr(), B1, S1,
                                               B2, S2, D, L.dtype_code(x), L.stream_ptr()), "vmr_ln_dwconv_fwd2")
            r += B1 * S1 + B2 * S2
        ctx.save_for_backward(x, gamma, beta, w2, mean, rstd)
        ctx.meta = (segs, tuple(w.shape))
        ctx.params = (gamma, beta, w)
        if tee:
            return u, x.view_as(x)
        return u

    @staticmethod
    def backward(ctx, du, dres=None):
        x, gamma, beta, w2, mean, rstd = ctx.saved_tensors
        segs, wshape = ctx.meta
        rows, D = x.shape
        du = du.contiguous()
        dres = None if dres is None else dres.contiguous()
        lib, st, dt = L.lib(), L.stream_ptr(), L.dtype_code(x)
        dn = torch.empty_like(x)
        mg, mb, mw = (main_grad(p_) for p_ in ctx.params)
        direct = mg is not None and mb is not None and mw is not None
        dw = mw if direct else torch.zeros(D, 7, device=x.device, dtype=torch.float32)
        r = 0
        defer = direct and DEFER_COLREDUCE and ctx.state is not None
        dx = torch.empty_like(x)
        dg = mg if direct else torch.zeros(D, device=x.device, dtype=torch.float32)
        db = mb if direct else torch.zeros_like(dg)
        if defer:
            # partial rows of both kernels stay in their own workspaces until the end-of-backward batched reduction
            nb = C.c_int32(0)
            for i in range(0, len(segs), 2):
                (B1, S1), (B2, S2) = segs[i], (segs[i + 1] if i + 1 < len(segs) else (0, 0))
                wsc = torch.empty((B1 * ((S1 + 63) // 64) + B2 * ((S2 + 63) // 64)) * D * 7, device=x.device,
                                  dtype=torch.float32)
                L.check(lib.vmr_dwconv_bwd2_deferred(du[r:].data_ptr(), x[r:].data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                                     mean[r:].data_ptr(), rstd[r:].data_ptr(), w2.data_ptr(),
                                                     dn[r:].data_ptr(), wsc.data_ptr(), B1, S1, B2, S2, D, dt, C.byref(nb), st),
                        "vmr_dwconv_bwd2_deferred")
                ctx.state.defer_colreduce(wsc, dw, dw, nb.value, 7 * D, 0, 0)
                r += B1 * S1 + B2 * S2
            wsl = torch.empty(L.ln_bwd_ws_floats(rows, D), device=x.device, dtype=torch.float32)
            L.check(lib.vmr_layernorm_bwd_deferred(dn.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                                   rstd.data_ptr(), _ptr(dres), dx.data_ptr(), None, wsl.data_ptr(), 0, rows, D,
                                                   dt, 0.0, 0, None, C.byref(nb), st), "vmr_layernorm_bwd_deferred")
            ctx.state.defer_colreduce(wsl, dg, db, nb.value, D, D, 512 if D <= 512 else (1024 if D <= 1024 else 2048))
            return dx, None, None, None, None, None, None, None
        ws = torch.empty(max(sum(b * ((sq + 63) // 64) for b, sq in segs) * D * 7, L.ln_bwd_ws_floats(rows, D)),
                         device=x.device,
                         dtype=torch.float32)
        for i in range(0, len(segs), 2):
            (B1, S1), (B2, S2) = segs[i], (segs[i + 1] if i + 1 < len(segs) else (0, 0))
            L.check(lib.vmr_dwconv_bwd2(du[r:].data_ptr(), x[r:].data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                        mean[r:].data_ptr(), rstd[r:].data_ptr(), w2.data_ptr(), dn[r:].data_ptr(),
                                        dw.data_ptr(), ws.data_ptr(), B1, S1, B2, S2, D, dt, st), "vmr_dwconv_bwd2")
            r += B1 * S1 + B2 * S2
        L.check(lib.vmr_layernorm_bwd(dn.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                      _ptr(dres), dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(), 0,
                                      rows, D, dt, 0.0, 0, None, st), "vmr_layernorm_bwd")
        if direct:
            return dx, None, None, None, None, None, None, None
        return dx, dg, db, dw.reshape(wshape), None, None, None, None


def ln_dwconv(x, gamma, beta, w, eps, segs, tee=False, cache=None):
    """cache: the model's WeightCache -- its PassState batches the parameter-gradient reductions of a backward pass;
    None = reduce right away."""
    return _LnDwConv.apply(x, gamma, beta, w, eps, tuple(segs), tee, cache)


# ---------------------------------------------------------------------------
# cosine similarity of one unit vector per clip against every row of [B, C, D] (BAN ContrastLoss)
# ---------------------------------------------------------------------------
class _CosRows(torch.autograd.Function):
    """sim [B, C] fp32 = <q_b, y_bc> / (max(|y_bc|, 1e-30) (1 + 1e-8)); q fp32 [B, D] unit rows, y [B, C, D] in the compute
    dtype (reference models/BANlib/model.py:639-671).  One pass over y each way (csrc/cosine.hip)."""

    @staticmethod
    def forward(ctx, q, y):
        L.require_gpu(q, y)
        q = q.float().contiguous()
        y = y.contiguous()
        B, Cn, D = y.shape
        sim = torch.empty(B, Cn, device=y.device, dtype=torch.float32)
        rn = torch.empty_like(sim)
        L.check(L.lib().vmr_cos_rows_fwd(q.data_ptr(), y.data_ptr(), sim.data_ptr(), rn.data_ptr(), B, Cn, D, L.dtype_code(y),
                                         L.stream_ptr()), "vmr_cos_rows_fwd")
        ctx.save_for_backward(q, y, sim, rn)
        return sim

    @staticmethod
    def backward(ctx, dsim):
        q, y, sim, rn = ctx.saved_tensors
        B, Cn, D = y.shape
        dy = torch.empty_like(y)
        dq = torch.zeros_like(q)
        L.check(L.lib().vmr_cos_rows_bwd(q.data_ptr(), y.data_ptr(), sim.data_ptr(), rn.data_ptr(), dsim.float().contiguous().data_ptr(),
                                         dy.data_ptr(), dq.data_ptr(), B, Cn, D, L.dtype_code(y), L.stream_ptr()), "vmr_cos_rows_bwd")
        return dq, dy


def cos_rows_supported(y) -> bool:
    return y.is_cuda and y.dim() == 3 and bool(L.lib().vmr_cos_rows_supported(int(y.shape[2])))


def cos_rows(q, y):
    return _CosRows.apply(q, y)


# ---------------------------------------------------------------------------
# the whole DepthwiseSeparableConvBlock as ONE autograd node (fused row-kernel backward)
# ---------------------------------------------------------------------------
FUSED_CONV_BLOCK = os.environ.get("VMR_FUSED_CONVBLOCK", "1") != "0"


def conv_block_fusable(x, layers) -> bool:
    """True when `conv_block` can take the one-node path: 16-bit activations of a width the fused backward kernel is
    built for, every parameter's gradient slot in the flat arena (so all parameter gradients accumulate in place and
    their column reductions are deferred), and K-major weight copies for the dX products."""
    if not (FUSED_CONV_BLOCK and DEFER_COLREDUCE and MERGE_DX_DW and USE_SLABS and AUX_BITS and DW_SIDE_STREAM is None):
        return False
    if not (x.is_cuda and x.dim() == 2 and L.is_16bit(x.dtype) and torch.is_grad_enabled()):
        return False
    D = x.shape[1]
    if not L.lib().vmr_convblock_bwd_supported(D, L.dtype_code(x)):
        return False
    for (g, b, dw, pw, pb) in layers:
        if any(main_grad(p) is None for p in (g, b, dw, pw, pb)):
            return False
        if tuple(pw.shape[:2]) != (D, D) or WeightCache.get_t([pw], x.dtype) is None:
            return False
    # the pointwise products must land on a kernel whose epilogue writes the ReLU / dropout mask as bits
    probe = L.GemmDesc()
    probe.lda = probe.ldb = probe.ldc = probe.ldr = D
    probe.M, probe.N, probe.K, probe.transA, probe.transB = x.shape[0], D, D, 0, 0
    probe.dtype, probe.Z1, probe.Z2, probe.splitk = L.dtype_code(x), 1, 1, 1
    probe.flags = L.EPI_BIAS | L.EPI_RELU | L.EPI_RESIDUAL
    p0 = layers[0]
    probe.A = probe.C = probe.residual = x.data_ptr()
    probe.B = p0[3].data_ptr()
    probe.bias = p0[4].data_ptr()
    return bool(L.lib().vmr_gemm_aux_bits_supported(C.byref(probe)))


class _ConvBlock(torch.autograd.Function):
    """n x { u = dw7(LN(x)); x <- drop(relu(u.W^T + b)) + x }  (reference DepthwiseSeparableConvBlock, layers.py:126-148)
    as ONE autograd node.  Forward = the same two launches per layer as `ln_dwconv` + `linear` (LN + dw conv, then
    the product with its bias / ReLU / dropout / residual / bit-mask epilogue).  Backward per layer: ONE merged
    dX + dW launch (the bias gradient rides on the dW product, vmr_gemm_t.a_colsum) and ONE vmr_convblock_bwd launch
    (conv backward + LayerNorm backward + residual gradient + the LOWER layer's ReLU / dropout mask) -- the dn, dy
    round trips and the relu_bwd_bias launch of every layer but the top one are gone.  Needs `conv_block_fusable`."""

    @staticmethod
    def forward(ctx, x, cache, segs, eps, drops, *params):
        L.require_gpu(x)
        x = x.contiguous()
        rows, D = x.shape
        assert rows == sum(b * s for b, s in segs) and len(segs) <= 2 and len(params) % 5 == 0
        nl = len(params) // 5
        lib, st, dt = L.lib(), L.stream_ptr(), L.dtype_code(x)
        (B1, S1), (B2, S2) = segs[0], (segs[1] if len(segs) > 1 else (0, 0))
        saved, scales = [], []
        for l in range(nl):
            g, b, dw, pw, pb = params[5 * l:5 * l + 5]
            drop = drops[l]
            u = torch.empty_like(x)
            mean = torch.empty(rows, device=x.device, dtype=torch.float32)
            rstd = torch.empty_like(mean)
            w2 = dw.detach().reshape(D, 7)
            L.check(lib.vmr_ln_dwconv_fwd2(x.data_ptr(), g.data_ptr(), b.data_ptr(), eps, w2.data_ptr(), u.data_ptr(),
                                           mean.data_ptr(), rstd.data_ptr(), B1, S1, B2, S2, D, dt, st), "vmr_ln_dwconv_fwd2")
            W = cache.get([pw], x.dtype, D)
            y = torch.empty_like(x)
            bits = torch.empty(rows, D // 8, device=x.device, dtype=torch.uint8)
            flags = L.EPI_BIAS | L.EPI_RELU | L.EPI_RESIDUAL | L.EPI_AUX | L.EPI_AUX_BITS | (L.EPI_DROPOUT if drop[0] > 0 else 0)
            gemm(u, W, y, rows, D, D, 0, 0, D, W.stride(0), D, dtype=dt, flags=flags, bias=pb, residual=x, aux=bits, ldr=D,
                 drop=drop)
            saved += [x, u, mean, rstd, bits]
            scales.append(1.0 / (1.0 - drop[0]) if drop[0] > 0 else 1.0)
            x = y
        ctx.save_for_backward(*saved)
        ctx.params, ctx.meta, ctx.state = params, (segs, nl, scales), cache.state
        return x

    @staticmethod
    def backward(ctx, dy):
        segs, nl, scales = ctx.meta
        saved, params, state = ctx.saved_tensors, ctx.params, ctx.state
        lib, st = L.lib(), L.stream_ptr()
        dy = dy.contiguous()
        rows, D = dy.shape
        dt = L.dtype_code(dy)
        (B1, S1), (B2, S2) = segs[0], (segs[1] if len(segs) > 1 else (0, 0))
        nbmax = lib.vmr_convblock_bwd_blocks(B1, S1, B2, S2, D)
        # top layer: its mask is applied here (no fused kernel above it); the bias gradient rides on the dW product
        dz = torch.empty_like(dy)
        L.check(lib.vmr_relu_bwd_bias(3, dy.data_ptr(), saved[5 * (nl - 1) + 4].data_ptr(), dz.data_ptr(), None, rows, D, D,
                                      scales[nl - 1], dt, 0.0, 0, None, None, 1.0, st), "vmr_relu_bwd_bias")
        for l in range(nl - 1, -1, -1):
            x, u, mean, rstd, _ = saved[5 * l:5 * l + 5]
            g, b, dw, pw, pb = params[5 * l:5 * l + 5]
            # dX = dz . Wt^T held back, launched with dW = dz^T . u (split-K slabs; + the previous layer's slab reduction)
            Wt = WeightCache.get_t([pw], dz.dtype)
            du, held = mm(dz, Wt, 0, 0, hold=True)
            merged = _cdiv(rows, 160) * _cdiv(D, 128) <= 512
            sk = splitk_for(D, D, rows, SPLITK_TARGET_MERGED if merged else 0)
            wgrad = main_grad(pw).view(D, D)
            if sk > 1:
                ws = torch.empty(sk, D, D, device=dy.device, dtype=torch.float32)
                gemm(dz, u, ws, D, D, rows, 1, 1, D, D, D, dtype=dt, flags=L.EPI_SLAB, splitk=sk, a_colsum=main_grad(pb),
                     held=held, state=state)
                state.reduce_later(ws, wgrad, sk, D * D, D, D)
            else:
                gemm(dz, u, wgrad, D, D, rows, 1, 1, D, D, D, dtype=dt, flags=L.EPI_ACCUM, splitk=1, a_colsum=main_grad(pb),
                     held=held)
            dx = torch.empty_like(dy)
            dzn = torch.empty_like(dy) if l > 0 else None
            part_dw = torch.empty(nbmax, 7 * D, device=dy.device, dtype=torch.float32)
            part_gb = torch.empty(nbmax, 2 * D, device=dy.device, dtype=torch.float32)
            nb = C.c_int32(0)
            L.check(lib.vmr_convblock_bwd(du.data_ptr(), x.data_ptr(), dy.data_ptr(),
                                          saved[5 * (l - 1) + 4].data_ptr() if l > 0 else None,
                                          scales[l - 1] if l > 0 else 1.0, g.data_ptr(), b.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), dw.detach().reshape(D, 7).data_ptr(), dx.data_ptr(), _ptr(dzn),
                                          part_dw.data_ptr(), part_gb.data_ptr(), B1, S1, B2, S2, D, dt, C.byref(nb), st),
                    "vmr_convblock_bwd")
            mdw = main_grad(dw).view(D * 7)
            state.defer_colreduce(part_dw, mdw, mdw, nb.value, 7 * D, 0, 0)
            state.defer_colreduce(part_gb, main_grad(g), main_grad(b), nb.value, D, D, 0)
            dy, dz = dx, dzn
        return (dy, None, None, None, None) + (None,) * len(params)


def conv_block(x, cache, segs, eps, drops, layers):
    """layers: per layer (LN gamma, LN beta, depthwise weight [D,1,7], pointwise weight [D,D,1], pointwise bias);
    drops: per layer dropout triple (DropCtx.next).  Call only when conv_block_fusable(x, layers)."""
    flat = [p for lay in layers for p in lay]
    return _ConvBlock.apply(x, cache, tuple(segs), eps, tuple(drops), *flat)


# ---------------------------------------------------------------------------
# attention cores: batched MFMA GEMMs + the masked softmax kernel
# ---------------------------------------------------------------------------
def _softmax_fwd(S, R, Cc, ldP, rmask, cmask, mode, H, cm_stride, scale, dtype, drop):
    """S: fp32 [Z1,Z2,R,ldS] -> (P dropped, Pkeep) in `dtype` [Z1,Z2,R,ldP]."""
    Z = S.shape[0] * S.shape[1]
    P = torch.empty(S.shape[0], S.shape[1], R, ldP, device=S.device, dtype=dtype)
    Pk = torch.empty_like(P) if drop[0] > 0 else None
    L.check(L.lib().vmr_softmax_fwd(S.data_ptr(), P.data_ptr(), _ptr(Pk), _ptr(rmask), cmask.data_ptr(), mode, Z, H,
                                    R, Cc, S.shape[3], ldP, cm_stride, scale, L.dtype_code(P), drop[0], drop[1],
                                    _ptr(drop[2]), L.stream_ptr()), "vmr_softmax_fwd")
    return P, (Pk if Pk is not None else P)


def _softmax_bwd(dP, Pk, R, Cc, scale, drop):
    Z = dP.shape[0] * dP.shape[1]
    dS = torch.empty_like(Pk)
    L.check(L.lib().vmr_softmax_bwd(dP.data_ptr(), Pk.data_ptr(), dS.data_ptr(), Z, R, Cc, dP.shape[3], Pk.shape[3],
                                    scale, L.dtype_code(Pk), drop[0], drop[1], _ptr(drop[2]), L.stream_ptr()),
            "vmr_softmax_bwd")
    return dS


def _attend_fwd(q4, k4, v4, o4, rmask, cmask, mode, H, cm_stride, scale, drop):
    """o = softmax(q.k^T*scale + mask).v over 4-D strided views [Z1,Z2,rows,hd]."""
    Z1, Z2, R, hd = q4.shape
    Ck = k4.shape[2]
    ld = _rup(Ck, 8)
    if FUSED_ATTENTION and L.lib().vmr_attention_fwd_supported(hd, Ck, L.dtype_code(q4)) and all(
            t.stride(3) == 1 and all(s % 8 == 0 for s in t.stride()[:3]) for t in (q4, k4, v4, o4)):
        # one kernel: scores + mask + softmax + dropout + context (csrc/attention.hip)
        P = torch.empty(Z1, Z2, R, ld, device=q4.device, dtype=q4.dtype)
        Pk = torch.empty_like(P) if drop[0] > 0 else None
        strides = (C.c_int64 * 12)(*[s for t in (q4, k4, v4, o4) for s in t.stride()[:3]])
        L.check(L.lib().vmr_attention_fwd(q4.data_ptr(), k4.data_ptr(), v4.data_ptr(), o4.data_ptr(), P.data_ptr(),
                                          _ptr(Pk), strides, _ptr(rmask), cmask.data_ptr(), mode, Z1, Z2, H, R, Ck, hd,
                                          ld, cm_stride, scale, L.dtype_code(q4), drop[0], drop[1], _ptr(drop[2]),
                                          L.stream_ptr()), "vmr_attention_fwd")
        return P, (Pk if Pk is not None else P)
    S = torch.empty(Z1, Z2, R, ld, device=q4.device, dtype=torch.float32)
    bmm4(q4, k4, S[..., :Ck], 0, 0)
    P, Pk = _softmax_fwd(S, R, Ck, ld, rmask, cmask, mode, H, cm_stride, scale, q4.dtype, drop)
    bmm4(P[..., :Ck], v4, o4, 0, 1)
    return P, Pk


def _attend_bwd(do4, q4, k4, v4, P, Pk, dq4, dk4, dv4, scale, drop, accumulate_dq):
    Z1, Z2, R, hd = q4.shape
    Ck = k4.shape[2]
    ld = P.shape[3]
    ts = (q4, k4, v4, do4, dq4, dk4, dv4)
    if FUSED_ATTENTION_BWD and L.lib().vmr_attention_bwd_supported(hd, R, Ck, L.dtype_code(q4)) and all(
            t.stride(3) == 1 and all(s % 8 == 0 for s in t.stride()[:3]) for t in ts):
        # one kernel per slice: dP, softmax/dropout backward, dQ, dK, dV (csrc/attention_bwd.hip)
        strides = (C.c_int64 * 21)(*[s for t in ts for s in t.stride()[:3]])
        L.check(L.lib().vmr_attention_bwd(do4.data_ptr(), q4.data_ptr(), k4.data_ptr(), v4.data_ptr(), Pk.data_ptr(),
                                          dq4.data_ptr(), dk4.data_ptr(), dv4.data_ptr(), strides, Z1, Z2, R, Ck, hd, ld,
                                          scale, 1 if accumulate_dq else 0, L.dtype_code(q4), drop[0], drop[1],
                                          _ptr(drop[2]), L.stream_ptr()), "vmr_attention_bwd")
        return
    bmm4(P[..., :Ck], do4, dv4, 1, 1)                                   # dV = P^T . dO
    dP = torch.empty(Z1, Z2, R, ld, device=q4.device, dtype=torch.float32)
    bmm4(do4, v4, dP[..., :Ck], 0, 0)                                   # dP = dO . V^T
    dS = _softmax_bwd(dP, Pk, R, Ck, scale, drop)
    if accumulate_dq:                                                   # dQ += dS . K (in place)
        bmm4(dS[..., :Ck], k4, dq4, 0, 1, flags=L.EPI_RESIDUAL, residual=dq4, ldr=dq4.stride(2))
    else:
        bmm4(dS[..., :Ck], k4, dq4, 0, 1)
    bmm4(dS[..., :Ck], q4, dk4, 1, 1)                                   # dK = dS^T . Q


class _DualAttention(torch.autograd.Function):
    """Self + cross attention cores of DualMultiAttention (reference
    models/layers.py:346-367) for BOTH directions of a DualAttentionBlock at once.
    Token matrices are packed [B*T video rows | B*L query rows]:
      qkv [N,3D] = (query | f_key | f_value) of every token as a `from` token,
      kv  [N,2D] = (t_key | t_value)         of every token as a `to` token.
    Direction v: from = video rows, to = query rows; direction t: the reverse.
    Returns the head-merged contexts (self, cross) [N, D] (pre s_dense/x_dense)."""

    @staticmethod
    def _views(qkv, kv, so, xo, B, T, Lq, H):
        D = qkv.shape[1] // 3
        hd = D // H
        Nv = B * T
        out = []
        for (r0, Lf, t0, Lt) in ((0, T, Nv, Lq), (Nv, Lq, 0, T)):
            f5 = qkv[r0:r0 + B * Lf].view(B, Lf, 3, H, hd)
            t5 = kv[t0:t0 + B * Lt].view(B, Lt, 2, H, hd)
            q4, kf4, vf4 = (f5[:, :, i].permute(0, 2, 1, 3) for i in range(3))
            kt4, vt4 = (t5[:, :, i].permute(0, 2, 1, 3) for i in range(2))
            so4 = so[r0:r0 + B * Lf].view(B, Lf, H, hd).permute(0, 2, 1, 3)
            xo4 = xo[r0:r0 + B * Lf].view(B, Lf, H, hd).permute(0, 2, 1, 3)
            out.append((q4, kf4, vf4, kt4, vt4, so4, xo4))
        return out

    @staticmethod
    def forward(ctx, qkv, kv, vmask, tmask, B, T, Lq, H, drops):
        L.require_gpu(qkv, kv)
        D = qkv.shape[1] // 3
        scale = 1.0 / math.sqrt(float(D // H))
        so = torch.empty(qkv.shape[0], D, device=qkv.device, dtype=qkv.dtype)
        xo = torch.empty_like(so)
        saved = []
        masks = ((vmask, tmask), (tmask, vmask))
        for d, (q4, kf4, vf4, kt4, vt4, so4, xo4) in enumerate(_DualAttention._views(qkv, kv, so, xo, B, T, Lq, H)):
            fm, tm = masks[d]
            saved += _attend_fwd(q4, kf4, vf4, so4, fm, fm, 0, H, 0, scale, drops[2 * d])
            saved += _attend_fwd(q4, kt4, vt4, xo4, fm, tm, 0, H, 0, scale, drops[2 * d + 1])
        ctx.save_for_backward(qkv, kv, *saved)
        ctx.meta = (B, T, Lq, H, scale, drops)
        return so, xo

    @staticmethod
    def backward(ctx, dso, dxo):
        qkv, kv, *saved = ctx.saved_tensors
        B, T, Lq, H, scale, drops = ctx.meta
        dso, dxo = dso.contiguous(), dxo.contiguous()
        dqkv = torch.empty_like(qkv)
        dkv = torch.empty_like(kv)
        fw = _DualAttention._views(qkv, kv, dso, dxo, B, T, Lq, H)
        bw = _DualAttention._views(dqkv, dkv, dso, dxo, B, T, Lq, H)
        for d in range(2):
            q4, kf4, vf4, kt4, vt4, dso4, dxo4 = fw[d]
            dq4, dkf4, dvf4, dkt4, dvt4, _, _ = bw[d]
            Ps, Pks, Px, Pkx = saved[4 * d:4 * d + 4]
            _attend_bwd(dso4, q4, kf4, vf4, Ps, Pks, dq4, dkf4, dvf4, scale, drops[2 * d], False)
            _attend_bwd(dxo4, q4, kt4, vt4, Px, Pkx, dq4, dkt4, dvt4, scale, drops[2 * d + 1], True)
        return dqkv, dkv, None, None, None, None, None, None, None


def dual_attention(qkv, kv, vmask, tmask, B, T, Lq, H, drops=(NO_DROP,) * 4):
    return _DualAttention.apply(qkv, kv, vmask.contiguous(), tmask.contiguous(), B, T, Lq, H, tuple(drops))


class _BatchAxisAttention(torch.autograd.Function):
    """Attention core of TopSelfAttention2 (reference models/layers.py:567-574):
    nn.MultiheadAttention without batch_first on [B,T,D] => for every time index t
    and head h, sample b attends over the samples b' of the batch; the float
    key_padding_mask vmask.T is ADDED to the logits.  qkv: [B*T, 3D] (in_proj
    output, rows b*T+t); returns the head-merged context [B*T, D] (pre out_proj)."""

    @staticmethod
    def forward(ctx, qkv, vmask, B, T, H, drop):
        L.require_gpu(qkv)
        D = qkv.shape[1] // 3
        hd = D // H
        scale = 1.0 / math.sqrt(float(hd))
        q4, k4, v4 = (qkv.view(B, T, 3, H, hd)[:, :, i].permute(1, 2, 0, 3) for i in range(3))  # [T,H,B,hd]
        o = torch.empty(B * T, D, device=qkv.device, dtype=qkv.dtype)
        o4 = o.view(B, T, H, hd).permute(1, 2, 0, 3)
        P, Pk = _attend_fwd(q4, k4, v4, o4, None, vmask, 1, H, T, scale, drop)
        ctx.save_for_backward(qkv, P, Pk)
        ctx.meta = (B, T, H, scale, drop)
        return o

    @staticmethod
    def backward(ctx, do):
        qkv, P, Pk = ctx.saved_tensors
        B, T, H, scale, drop = ctx.meta
        D = qkv.shape[1] // 3
        hd = D // H
        do = do.contiguous()
        dqkv = torch.empty_like(qkv)
        q4, k4, v4 = (qkv.view(B, T, 3, H, hd)[:, :, i].permute(1, 2, 0, 3) for i in range(3))
        dq4, dk4, dv4 = (dqkv.view(B, T, 3, H, hd)[:, :, i].permute(1, 2, 0, 3) for i in range(3))
        do4 = do.view(B, T, H, hd).permute(1, 2, 0, 3)
        _attend_bwd(do4, q4, k4, v4, P, Pk, dq4, dk4, dv4, scale, drop, False)
        return dqkv, None, None, None, None, None


def batch_axis_attention(qkv, vmask, B, T, H, drop=NO_DROP):
    return _BatchAxisAttention.apply(qkv, vmask.contiguous(), B, T, H, drop)


# ---------------------------------------------------------------------------
# generic differentiable batched product (CQAttention contractions)
# ---------------------------------------------------------------------------
_BMM_FORMS = {(0, 0), (0, 1), (1, 1)}


def _bmm_raw(a, b, ta, tb, out_f32=False):
    """a,b: [Z, r, c] contiguous -> [Z, M, N]"""
    Z = a.shape[0]
    M = a.shape[2] if ta else a.shape[1]
    N = b.shape[2] if tb else b.shape[1]
    Np = _rup(N, 8)
    K = a.shape[1] if ta else a.shape[2]
    tiles = _cdiv(M, 128) * _cdiv(N, 128) * Z
    if tiles < 256 and K >= 512 and a.dtype != torch.float32:
        # a handful of tiles with a long K (the [Lc x Lq] score / score-gradient products of CQAttention, K = D):
        # split K so the grid fills the chip; fp32 partials meet through atomics (<= 8 adders per address)
        sk = max(2, min(8, K // 128, _cdiv(512, tiles)))
        cbuf = torch.zeros(Z, M, Np, device=a.device, dtype=torch.float32)
        c = cbuf[..., :N]
        bmm4(a.unsqueeze(0), b.unsqueeze(0), c.unsqueeze(0), ta, tb, flags=L.EPI_ACCUM, splitk=sk)
        return c if out_f32 else cbuf.to(a.dtype)[..., :N]
    cbuf = torch.empty(Z, M, Np, device=a.device, dtype=torch.float32 if out_f32 else a.dtype)
    c = cbuf[..., :N]
    bmm4(a.unsqueeze(0), b.unsqueeze(0), c.unsqueeze(0), ta, tb)
    return c


class _Bmm(torch.autograd.Function):
    """c[z] = op(a[z]) . op(b[z]) on the MFMA GEMM (reference torch.matmul sites of
    CQAttention, models/layers.py:422-423,435).  Forms: (ta,tb) = (0,0): a[M,K].b[N,K]^T;
    (0,1): a[M,K].b[K,N]; (1,1): a[K,M]^T.b[K,N]."""

    @staticmethod
    def forward(ctx, a, b, ta, tb, out_f32):
        L.require_gpu(a, b)
        assert (ta, tb) in _BMM_FORMS
        a = a if a.stride(-1) == 1 else a.contiguous()
        b = b if b.stride(-1) == 1 else b.contiguous()
        ctx.save_for_backward(a, b)
        ctx.meta = (ta, tb)
        return _bmm_raw(a, b, ta, tb, out_f32)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        ta, tb = ctx.meta
        dc = dc.to(a.dtype)
        if dc.stride(-1) != 1 or dc.stride(-2) % 8 or dc.stride(0) % 8:
            dcp = torch.zeros(dc.shape[0], dc.shape[1], _rup(dc.shape[2], 8), device=dc.device, dtype=dc.dtype)
            dcp[..., :dc.shape[2]] = dc
            dc = dcp[..., :dc.shape[2]]
        if (ta, tb) == (0, 0):
            da, db = _bmm_raw(dc, b, 0, 1), _bmm_raw(dc, a, 1, 1)
        elif (ta, tb) == (0, 1):
            da, db = _bmm_raw(dc, b, 0, 0), _bmm_raw(a, dc, 1, 1)
        else:
            da, db = _bmm_raw(b, dc, 0, 0), _bmm_raw(a, dc, 0, 1)
        return da, db, None, None, None


def bmm(a, b, ta, tb, out_f32=False):
    return _Bmm.apply(a, b, ta, tb, out_f32)


class _Dropout(torch.autograd.Function):
    """Stand-alone inverted dropout (the score-path dropout of CQAttention,
    reference models/layers.py:431-432); mask regenerated in the backward.  tee=True also returns an alias of x: hand
    it to x's OTHER consumer, whose gradient is then added inside this backward's kernel (no separate add pass)."""

    @staticmethod
    def forward(ctx, x, drop, tee=False):
        L.require_gpu(x)
        x = x.contiguous()
        D = x.shape[-1]
        rows = x.numel() // D
        y = torch.empty_like(x)
        L.check(L.lib().vmr_cast(x.data_ptr(), L.dtype_code(x), y.data_ptr(), L.dtype_code(y), rows, D, D, D,
                                 drop[0], drop[1], _ptr(drop[2]), L.stream_ptr()), "vmr_cast")
        ctx.drop = drop
        if tee:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dxtra=None):
        drop = ctx.drop
        if dy is None:
            return dxtra, None, None
        dy = dy.contiguous()
        D = dy.shape[-1]
        rows = dy.numel() // D
        dx = torch.empty_like(dy)
        if dxtra is not None:
            dxtra = dxtra.contiguous()
            assert dxtra.shape == dy.shape and dxtra.dtype == dy.dtype
        L.check(L.lib().vmr_relu_bwd_bias(4 if dxtra is not None else 2, dy.data_ptr(), _ptr(dxtra), dx.data_ptr(), None, rows, D,
                                          D, 1.0 / (1.0 - drop[0]), L.dtype_code(dy), drop[0], drop[1], _ptr(drop[2]),
                                          None, 1.0, L.stream_ptr()), "vmr_relu_bwd_bias")
        return dx, None, None


class _SplitRows(torch.autograd.Function):
    """(X[:n0], X[n0:]) of a packed token matrix; backward = ONE concat of the two gradients (autograd's slice backward
    zero-fills the full matrix twice, copies each slice in and adds the two)."""

    @staticmethod
    def forward(ctx, X, n0):
        ctx.n0, ctx.shape = n0, tuple(X.shape)
        ctx.meta = (X.dtype, X.device)
        return X[:n0], X[n0:]

    @staticmethod
    def backward(ctx, da, db):
        n0, shape = ctx.n0, ctx.shape
        if da is None:
            da = torch.zeros((n0,) + shape[1:], dtype=ctx.meta[0], device=ctx.meta[1])
        if db is None:
            db = torch.zeros((shape[0] - n0,) + shape[1:], dtype=ctx.meta[0], device=ctx.meta[1])
        return torch.cat([da.reshape((n0,) + shape[1:]), db.reshape((shape[0] - n0,) + shape[1:])], 0), None


def split_rows(X, n0):
    return _SplitRows.apply(X, n0)


def dropout(x, drop, tee=False):
    """tee=True: returns (dropout(x), alias of x) -- see _Dropout."""
    if drop[0] <= 0.0:
        return (x, x) if tee else x
    return _Dropout.apply(x, drop, tee)


# ---------------------------------------------------------------------------
# CQAttention: the two masked softmaxes of the trilinear score
# ---------------------------------------------------------------------------
class _CQSoftmax(torch.autograd.Function):
    """(S_row, S_col) = softmaxes over q / over c of S2 + rowterm + colterm with the -1e30 masks
    (reference models/layers.py:419-421).  S2: fp32 [B,Lc,>=Lq] view (last stride 1) from the
    batched GEMM; outputs [B,Lc,Lq] views of 8-padded buffers in `dtype`."""

    @staticmethod
    def forward(ctx, S2, rowterm, colterm, cmask, qmask, dtype):
        L.require_gpu(S2)
        B, Lc, Lq = S2.shape
        assert S2.stride(2) == 1 and S2.stride(0) == Lc * S2.stride(1)
        ldS, ldP = S2.stride(1), _rup(Lq, 8)
        Srow = torch.empty(B, Lc, ldP, device=S2.device, dtype=dtype)
        Scol = torch.empty_like(Srow)
        rt = None if rowterm is None else rowterm.contiguous().float()
        ct = None if colterm is None else colterm.contiguous().float()
        L.check(L.lib().vmr_cq_softmax_fwd(S2.data_ptr(), _ptr(rt), _ptr(ct), cmask.data_ptr(), qmask.data_ptr(),
                                           Srow.data_ptr(), Scol.data_ptr(), B, Lc, Lq, ldS, ldP, L.dtype_code(Srow),
                                           L.stream_ptr()), "vmr_cq_softmax_fwd")
        ctx.save_for_backward(Srow, Scol)
        ctx.meta = (B, Lc, Lq, ldP, rowterm is not None, colterm is not None,
                    None if rowterm is None else tuple(rowterm.shape), None if colterm is None else tuple(colterm.shape))
        return Srow[..., :Lq], Scol[..., :Lq]

    @staticmethod
    def backward(ctx, dSrow, dScol):
        Srow, Scol = ctx.saved_tensors
        B, Lc, Lq, ldP, has_r, has_c, rshape, cshape = ctx.meta
        dev = Srow.device

        def pad(g):   # gradients arrive as [B,Lc,Lq] (any strides): bring them to the padded layout
            buf = torch.zeros(B, Lc, ldP, device=dev, dtype=Srow.dtype)
            buf[..., :Lq] = g
            return buf
        dSr, dSc = pad(dSrow), pad(dScol)
        dS2 = torch.zeros(B, Lc, ldP, device=dev, dtype=torch.float32)
        drow = torch.empty(B, Lc, device=dev, dtype=torch.float32) if has_r else None
        dcol = torch.empty(B, Lq, device=dev, dtype=torch.float32) if has_c else None
        L.check(L.lib().vmr_cq_softmax_bwd(dSr.data_ptr(), dSc.data_ptr(), Srow.data_ptr(), Scol.data_ptr(),
                                           dS2.data_ptr(), _ptr(drow), _ptr(dcol), B, Lc, Lq, ldP, ldP,
                                           L.dtype_code(Srow), L.stream_ptr()), "vmr_cq_softmax_bwd")
        return (dS2[..., :Lq], None if drow is None else drow.reshape(rshape),
                None if dcol is None else dcol.reshape(cshape), None, None, None)


class _CQScore(torch.autograd.Function):
    """CQAttention similarity + both masked softmaxes in ONE kernel (csrc/cqscore.hip): the short-stream operand
    staged in LDS, the long stream read from HBM once, the score tile in registers.  Backward: the existing
    vmr_cq_softmax_bwd on the saved probabilities, then the two small batched products."""

    @staticmethod
    def forward(ctx, lng, short_op, shortterm, mask_long, mask_short, orient):
        L.require_gpu(lng, short_op)
        B, Ll, D = lng.shape
        Ls = short_op.shape[1]
        lng, short_op = lng.contiguous(), short_op.contiguous()
        rows, cols = (Ll, Ls) if orient == 0 else (Ls, Ll)
        ldP = _rup(cols, 8)
        Srow = torch.empty(B, rows, ldP, device=lng.device, dtype=lng.dtype)
        Scol = torch.empty_like(Srow)
        st = shortterm.contiguous().float()
        def launch():
            L.check(L.lib().vmr_cq_score_fwd(lng.data_ptr(), short_op.data_ptr(), st.data_ptr(), mask_long.data_ptr(),
                                             mask_short.data_ptr(), Srow.data_ptr(), Scol.data_ptr(), None, None, B, Ll, Ls,
                                             D, ldP, orient, L.dtype_code(lng), L.stream_ptr()), "vmr_cq_score_fwd")
        if CQ_HOOK is not None:
            CQ_HOOK(launch, B, Ll, Ls, D)
        else:
            launch()
        ctx.save_for_backward(lng, short_op, Srow, Scol)
        ctx.meta = (B, Ll, Ls, D, ldP, orient, tuple(shortterm.shape))
        return Srow[..., :cols], Scol[..., :cols]

    @staticmethod
    def backward(ctx, dSrow, dScol):
        lng, short_op, Srow, Scol = ctx.saved_tensors
        B, Ll, Ls, D, ldP, orient, tshape = ctx.meta
        rows, cols = (Ll, Ls) if orient == 0 else (Ls, Ll)
        dev = Srow.device

        def pad(g):
            buf = torch.zeros(B, rows, ldP, device=dev, dtype=Srow.dtype)
            buf[..., :cols] = g
            return buf
        dSr, dSc = pad(dSrow), pad(dScol)
        dS2 = torch.zeros(B, rows, ldP, device=dev, dtype=torch.float32)
        dterm = torch.empty(B, Ls, device=dev, dtype=torch.float32)
        # S2[c,q] in the stored layout: orient 0 -> c = long row, q = short row (colterm); orient 1 -> c = short row (rowterm)
        L.check(L.lib().vmr_cq_softmax_bwd(dSr.data_ptr(), dSc.data_ptr(), Srow.data_ptr(), Scol.data_ptr(), dS2.data_ptr(),
                                           dterm.data_ptr() if orient == 1 else None,
                                           dterm.data_ptr() if orient == 0 else None, B, rows, cols, ldP, ldP,
                                           L.dtype_code(Srow), L.stream_ptr()), "vmr_cq_softmax_bwd")
        dS = dS2.to(Srow.dtype)[..., :cols]                      # [B, rows, cols], row stride ldP
        if orient == 0:                                          # S2 = long . short_op^T
            dlong = _bmm_raw(dS, short_op, 0, 1)                 # [B,Ll,Ls] . [B,Ls,D]
            dshort = _bmm_raw(dS, lng, 1, 1)                     # [B,Ll,Ls]^T . [B,Ll,D]
        else:                                                    # S2 = short_op . long^T
            dshort = _bmm_raw(dS, lng, 0, 1)                     # [B,Ls,Ll] . [B,Ll,D]
            dlong = _bmm_raw(dS, short_op, 1, 1)                 # [B,Ls,Ll]^T . [B,Ls,D]
        return dlong, dshort, dterm.reshape(tshape), None, None, None


class _CQBlock(torch.autograd.Function):
    """The whole CQAttention core (reference models/layers.py:417-424) as fused launches:
    forward  = vmr_cq_score_fwd (trilinear similarity + both masked softmaxes, written as fp32 long-major rows) ->
               vmr_cq_apply_fwd (c2q, S_t^T.C, q2c and the 4-way concat, one workgroup per clip and 128-channel slice);
    backward = vmr_cq_apply_bwd (dctx, dqry and per-slice partials of dS_ / dS_t) -> vmr_cq_softmax_bwd_parts ->
               vmr_cq_score_bwd (gradients of the two score operands).
    The two probability matrices ([B, Ll, SP] fp32, 1.5 MB each at cfg2) are the only intermediates in HBM.
    Inputs: ctx [B,Lc,D], qry [B,Lq,D] (apply stage), lng / short_op (the score operands: the long stream, possibly a
    dropout copy, and the rank-1-folded short operand), shortterm [B,Ls]; orient 0: context is the long stream."""

    @staticmethod
    def forward(ctx_, ctx, qry, lng, short_op, shortterm, mask_long, mask_short, orient):
        L.require_gpu(ctx, qry, lng, short_op)
        B, Lc, D = ctx.shape
        Lq = qry.shape[1]
        ctx, qry, lng, short_op = ctx.contiguous(), qry.contiguous(), lng.contiguous(), short_op.contiguous()
        Ll, Ls = lng.shape[1], short_op.shape[1]
        assert (Ll, Ls) == ((Lc, Lq) if orient == 0 else (Lq, Lc))
        SP = _rup(Ls, 8)
        Pt = torch.empty(B, Ll, SP, device=ctx.device, dtype=torch.float32)      # softmax over the short index
        Pv = torch.empty_like(Pt)                                                # softmax over the long index
        st = shortterm.contiguous().float()
        lib, stream = L.lib(), L.stream_ptr()

        cstat = torch.empty(lib.vmr_cq_score_ws_floats(B), device=ctx.device, dtype=torch.float32)

        def launch():
            L.check(lib.vmr_cq_score_fwd_ws(lng.data_ptr(), short_op.data_ptr(), st.data_ptr(), mask_long.data_ptr(),
                                            mask_short.data_ptr(), None, None, Pt.data_ptr(), Pv.data_ptr(), cstat.data_ptr(),
                                            B, Ll, Ls, D, 0, orient, L.dtype_code(lng), stream), "vmr_cq_score_fwd_ws")
        if CQ_HOOK is not None:
            CQ_HOOK(launch, B, Ll, Ls, D)
        else:
            launch()
        # S_ = softmax over q, S_t = softmax over c: q is the short index when the context is the long stream
        S_, S_t = (Pt, Pv) if orient == 0 else (Pv, Pt)
        out = torch.empty(B * Lc, 4 * D, device=ctx.device, dtype=ctx.dtype)

        def apply_():
            L.check(lib.vmr_cq_apply_fwd(ctx.data_ptr(), qry.data_ptr(), S_.data_ptr(), S_t.data_ptr(), out.data_ptr(), B, Lc,
                                         Lq, D, L.dtype_code(ctx), stream), "vmr_cq_apply_fwd")
        if CQ_APPLY_HOOK is not None:
            CQ_APPLY_HOOK(apply_, "fwd", B, Lc, Lq, D)
        else:
            apply_()
        ctx_.save_for_backward(ctx, qry, lng, short_op, S_, S_t)
        ctx_.meta = (B, Lc, Lq, D, orient, tuple(shortterm.shape))
        return out

    @staticmethod
    def backward(ctx_, dout):
        ctx, qry, lng, short_op, S_, S_t = ctx_.saved_tensors
        B, Lc, Lq, D, orient, tshape = ctx_.meta
        dev, lib, stream = ctx.device, L.lib(), L.stream_ptr()
        dout = dout.contiguous()
        dctx, dqry = torch.empty_like(ctx), torch.empty_like(qry)
        parts = torch.empty(L.cq_apply_parts_floats(B, Lc, Lq, D), device=dev, dtype=torch.float32)

        def apply_():
            L.check(lib.vmr_cq_apply_bwd(dout.data_ptr(), ctx.data_ptr(), qry.data_ptr(), S_.data_ptr(), S_t.data_ptr(),
                                         dctx.data_ptr(), dqry.data_ptr(), parts.data_ptr(), B, Lc, Lq, D,
                                         L.dtype_code(ctx), stream), "vmr_cq_apply_bwd")
        if CQ_APPLY_HOOK is not None:
            CQ_APPLY_HOOK(apply_, "bwd", B, Lc, Lq, D)
        else:
            apply_()
        Ll, Ls = lng.shape[1], short_op.shape[1]
        dS = torch.empty_like(S_)                                   # fp32 long-major [B, Ll, SP]
        dterm = torch.empty(B, Ls, device=dev, dtype=torch.float32)
        L.check(lib.vmr_cq_softmax_bwd_parts(parts.data_ptr(), S_.data_ptr(), S_t.data_ptr(), dS.data_ptr(), dterm.data_ptr(),
                                             B, Lc, Lq, D, stream), "vmr_cq_softmax_bwd_parts")
        dlng, dshort = torch.empty_like(lng), torch.empty_like(short_op)
        L.check(lib.vmr_cq_score_bwd(lng.data_ptr(), short_op.data_ptr(), dS.data_ptr(), dlng.data_ptr(), dshort.data_ptr(), B,
                                     Ll, Ls, D, L.dtype_code(lng), stream), "vmr_cq_score_bwd")
        return dctx, dqry, dlng, dshort, dterm.reshape(tshape), None, None, None


CQ_APPLY_HOOK = None   # bench.py: callable(launch, "fwd"|"bwd", B, Lc, Lq, D) around the fused CQ apply kernels
FUSED_CQ_APPLY = os.environ.get("VMR_FUSED_CQ_APPLY", "1") != "0"   # csrc/cqapply.hip


def cq_block_supported(Lc, Lq, D, dtype):
    """Both halves of the fused CQAttention core take this shape (score kernel: long stream <= 128 rows)."""
    Ll, Ls = max(Lc, Lq), min(Lc, Lq)
    code = L.F32 if dtype == torch.float32 else L.BF16
    if Ll > 128:      # (BaseFast's T = 256: the score kernel's row-split form + the 64-channel-slice apply kernels)
        return bool(FUSED_CQ_APPLY and FUSED_CQ_SCORE and L.lib().vmr_cq_score_split_supported(Ll, Ls, D, code) and
                    L.lib().vmr_cq_apply_supported(Lc, Lq, D, code))
    return (FUSED_CQ_APPLY and cq_score_supported(Ll, Ls, D, dtype) and
            bool(L.lib().vmr_cq_apply_supported(Lc, Lq, D, code)))


def cq_block(ctx, qry, lng, short_op, shortterm, mask_long, mask_short, orient):
    """cat4 [B*Lc, 4D] of CQAttention from the context / query streams and the two score operands.  orient 0: the context
    is the long stream (Lq <= Lc), 1: the short one (Lq > Lc) -- the apply kernels derive the same from the lengths, so a
    tie is orient 0 (SeqPAN.cq_attention_core does exactly that)."""
    return _CQBlock.apply(ctx, qry, lng, short_op, shortterm, mask_long.contiguous(), mask_short.contiguous(), orient)


def cq_score_supported(Ll, Ls, D, dtype):
    return FUSED_CQ_SCORE and bool(L.lib().vmr_cq_score_supported(Ll, Ls, D, L.F32 if dtype == torch.float32 else L.BF16))


def cq_score(lng, short_op, shortterm, mask_long, mask_short, orient):
    """(S_row, S_col) of CQAttention from the long stream [B,Ll,D], the folded short operand [B,Ls,D] and its rank-1
    term [B,Ls]; orient 0: context = long stream, 1: context = short stream."""
    return _CQScore.apply(lng, short_op, shortterm, mask_long.contiguous(), mask_short.contiguous(), orient)


def cq_softmax(S2, rowterm, colterm, cmask, qmask, dtype):
    return _CQSoftmax.apply(S2, rowterm, colterm, cmask.contiguous(), qmask.contiguous(), dtype)


# ---------------------------------------------------------------------------
# fused elementwise programs (dual-attention gating, CQ concat)
# ---------------------------------------------------------------------------
def _elt(op, a, b, c, d, e, rowmask, outs, rows, D):
    o = [_ptr(t) for t in outs] + [None] * (4 - len(outs))
    L.check(L.lib().vmr_eltwise(op, _ptr(a), _ptr(b), _ptr(c), _ptr(d), _ptr(e), _ptr(rowmask), o[0], o[1], o[2], o[3],
                                rows, D, L.dtype_code(a), L.stream_ptr()), "vmr_eltwise")


class _CrossGate(torch.autograd.Function):
    """s_score * x_value + x_score * s_value (reference models/layers.py:374), one pass."""

    @staticmethod
    def forward(ctx, ss, sv, xs, xv):
        L.require_gpu(ss)
        ss, sv, xs, xv = (t.contiguous() for t in (ss, sv, xs, xv))
        out = torch.empty_like(ss)
        _elt(0, ss, sv, xs, xv, None, None, [out], ss.shape[0], ss.shape[1])
        ctx.save_for_backward(ss, sv, xs, xv)
        return out

    @staticmethod
    def backward(ctx, do):
        ss, sv, xs, xv = ctx.saved_tensors
        do = do.contiguous()
        g = [torch.empty_like(ss) for _ in range(4)]
        _elt(1, do, ss, sv, xs, xv, None, g, ss.shape[0], ss.shape[1])
        return tuple(g)


def cross_gate(ss, sv, xs, xv):
    return _CrossGate.apply(ss, sv, xs, xv)


class _SigmoidGate(torch.autograd.Function):
    """sigmoid(mask_logits(scores, row_mask)) * values on the [rows, 2D] = (scores | values) GEMM
    output (reference models/layers.py:380)."""

    @staticmethod
    def forward(ctx, sv2, rowmask):
        L.require_gpu(sv2)
        sv2 = sv2.contiguous()
        rows, D = sv2.shape[0], sv2.shape[1] // 2
        out = torch.empty(rows, D, device=sv2.device, dtype=sv2.dtype)
        _elt(2, sv2, None, None, None, None, rowmask, [out], rows, D)
        ctx.save_for_backward(sv2, rowmask)
        return out

    @staticmethod
    def backward(ctx, do):
        sv2, rowmask = ctx.saved_tensors
        do = do.contiguous()
        rows, D = do.shape
        dsv = torch.empty_like(sv2)
        _elt(3, do, sv2, None, None, None, rowmask, [dsv], rows, D)
        return dsv, None


def sigmoid_gate(sv2, rowmask_f32):
    return _SigmoidGate.apply(sv2, rowmask_f32)


class _Cat4(torch.autograd.Function):
    """[C, c2q, C*c2q, C*q2c] of CQAttention (reference models/layers.py:424) in one pass."""

    @staticmethod
    def forward(ctx, C_, c2q, q2c):
        L.require_gpu(C_)
        C_, c2q, q2c = (t.contiguous() for t in (C_, c2q, q2c))
        rows, D = C_.shape
        out = torch.empty(rows, 4 * D, device=C_.device, dtype=C_.dtype)
        _elt(4, C_, c2q, q2c, None, None, None, [out], rows, D)
        ctx.save_for_backward(C_, c2q, q2c)
        return out

    @staticmethod
    def backward(ctx, dcat):
        C_, c2q, q2c = ctx.saved_tensors
        dcat = dcat.contiguous()
        rows, D = C_.shape
        g = [torch.empty_like(C_) for _ in range(3)]
        _elt(5, dcat, C_, c2q, q2c, None, None, g, rows, D)
        return tuple(g)


def cat4(C_, c2q, q2c):
    return _Cat4.apply(C_, c2q, q2c)


# ---------------------------------------------------------------------------
# boundary-label cross-entropy (reference models/loss.py:43-54)
# ---------------------------------------------------------------------------
class _SoftCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, zs, ze, ys, ye):
        L.require_gpu(zs, ze, ys, ye)
        zs, ze, ys, ye = (t.contiguous().float() for t in (zs, ze, ys, ye))
        B, T = zs.shape
        loss = torch.zeros(1, device=zs.device, dtype=torch.float32)
        lse = torch.empty(2 * B, device=zs.device, dtype=torch.float32)
        L.check(L.lib().vmr_soft_ce_fwd(zs.data_ptr(), ze.data_ptr(), ys.data_ptr(), ye.data_ptr(), loss.data_ptr(),
                                        lse.data_ptr(), B, T, L.stream_ptr()), "vmr_soft_ce_fwd")
        ctx.save_for_backward(zs, ze, ys, ye, lse)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        zs, ze, ys, ye, lse = ctx.saved_tensors
        B, T = zs.shape
        g = g.reshape(1).contiguous().float()
        dzs, dze = torch.empty_like(zs), torch.empty_like(ze)
        L.check(L.lib().vmr_soft_ce_bwd(zs.data_ptr(), ze.data_ptr(), ys.data_ptr(), ye.data_ptr(), lse.data_ptr(),
                                        g.data_ptr(), dzs.data_ptr(), dze.data_ptr(), B, T, L.stream_ptr()),
                "vmr_soft_ce_bwd")
        return dzs, dze, None, None


def soft_ce(zs, ze, ys, ye):
    return _SoftCE.apply(zs, ze, ys, ye)


class _Embedding(torch.autograd.Function):
    """F.embedding(idx, table, padding_idx) as a HIP row gather; the backward is a float-atomic
    scatter-add (graph-capturable: no host-side segment counting as in torch's kernel)."""

    @staticmethod
    def forward(ctx, idx, table, padding_idx):
        L.require_gpu(idx, table)
        idx = idx.contiguous()
        table = table.contiguous().float()
        n, D = idx.numel(), table.shape[1]
        out = torch.empty(*idx.shape, D, device=table.device, dtype=torch.float32)
        L.check(L.lib().vmr_embedding_fwd(idx.data_ptr(), table.data_ptr(), out.data_ptr(), n, D, table.shape[0],
                                          L.stream_ptr()), "vmr_embedding_fwd")
        ctx.save_for_backward(idx)
        ctx.meta = (tuple(table.shape), padding_idx)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx,) = ctx.saved_tensors
        shape, padding_idx = ctx.meta
        dout = dout.contiguous().float()
        dtable = torch.zeros(shape, device=dout.device, dtype=torch.float32)
        L.check(L.lib().vmr_embedding_bwd(idx.data_ptr(), dout.data_ptr(), dtable.data_ptr(), idx.numel(), shape[1],
                                          shape[0], padding_idx, L.stream_ptr()), "vmr_embedding_bwd")
        return None, dtable, None


def embedding(idx, table, padding_idx=0):
    return _Embedding.apply(idx, table, padding_idx)


def dropout_mask(n: int, p: float, seed: int, device) -> torch.Tensor:
    """Materialise the counter-based dropout multiplier (tests / oracle injection)."""
    m = torch.empty(n, device=device, dtype=torch.float32)
    L.check(L.lib().vmr_dropout_mask(m.data_ptr(), n, p, seed, L.stream_ptr()), "vmr_dropout_mask")
    return m


# ---------------------------------------------------------------------------
# BAN 2-D proposal map (csrc/map2d.hip)
# ---------------------------------------------------------------------------
class Map2dLayout:
    """Cell layout of the reference's mask2d (models/BANlib/model.py:226-325): `grow[k]` = MaxPool1d kernel
    size - 1 of pooler k; cells in `maskij` order (main diagonal, then one diagonal per pooler)."""

    def __init__(self, N: int, pooling_counts=None, device=None):
        if pooling_counts is None:                       # DenseMaxPool: every diagonal, MaxPool1d(2, 1) each
            grow = [1] * (N - 1)
        else:                                            # SparseMaxPool / SparseBoundaryCat
            assert len(pooling_counts) <= 3, "the reference's 4th level (kernel 7, stride 8) does not run"
            grow = []
            for lvl, c in enumerate(pooling_counts):
                grow += [1 if lvl == 0 else 2 * lvl] * c
        self.N = N
        self.grow_host = np.asarray(grow, dtype=np.int32)
        off = np.cumsum(self.grow_host)
        assert len(off) == 0 or off[-1] < N, "pooling_counts reach past the sequence"
        ii, jj = [np.arange(N)], [np.arange(N)]
        for o in off:
            ii.append(np.arange(0, N - o)); jj.append(np.arange(o, N))
        self.ii, self.jj = np.concatenate(ii), np.concatenate(jj)
        self.C = int(self.ii.size)
        cell_of = -np.ones((N, N), dtype=np.int32)
        cell_of[self.ii, self.jj] = np.arange(self.C, dtype=np.int32)
        self.mask2d_host = cell_of >= 0
        self.device = None
        if device is not None:
            self.to(device)

    def to(self, device):
        if self.device != device:
            self.grow = torch.from_numpy(self.grow_host).to(device)
            cell_of = -np.ones((self.N, self.N), dtype=np.int32)
            cell_of[self.ii, self.jj] = np.arange(self.C, dtype=np.int32)
            self.cell_of = torch.from_numpy(cell_of).to(device)
            self.mask2d = torch.from_numpy(self.mask2d_host).to(device)
            self.ii_t = torch.from_numpy(self.ii).to(device)
            self.jj_t = torch.from_numpy(self.jj).to(device)
            self.device = device
        return self


class _Map2dPool(torch.autograd.Function):
    """(M, R): M[b,c] = max of x[b, i_c..j_c]; R[b,c] = ps[b,i_c] + pe[b,j_c] (compact cells)."""

    @staticmethod
    def forward(ctx, x, ps, pe, layout):
        L.require_gpu(x)
        B, N, F = x.shape
        assert N == layout.N
        x = x.contiguous()
        M = torch.empty(B, layout.C, F, device=x.device, dtype=x.dtype)
        R = None
        if ps is not None:
            ps, pe = ps.contiguous(), pe.contiguous()
            assert ps.shape == (B * N, F) and pe.shape == (B * N, F) and ps.dtype == x.dtype
            R = torch.empty_like(M)
        L.check(L.lib().vmr_map2d_pool_fwd(x.data_ptr(), _ptr(ps), _ptr(pe), F, layout.grow.data_ptr(),
                                           layout.grow_host.ctypes.data, len(layout.grow_host), M.data_ptr(), _ptr(R), B, N,
                                           F, L.dtype_code(x), L.stream_ptr()), "vmr_map2d_pool_fwd")
        ctx.save_for_backward(x)
        ctx.layout, ctx.has_p = layout, ps is not None
        return (M, R) if R is not None else M

    @staticmethod
    def backward(ctx, dM, dR=None):
        (x,) = ctx.saved_tensors
        layout = ctx.layout
        B, N, F = x.shape
        dM = dM.contiguous()
        dx = torch.empty_like(x)
        dps = dpe = None
        if ctx.has_p:
            dR = dR.contiguous()
            dps = torch.empty(B * N, F, device=x.device, dtype=x.dtype)
            dpe = torch.empty_like(dps)
        L.check(L.lib().vmr_map2d_pool_bwd(x.data_ptr(), dM.data_ptr(), _ptr(dR) if ctx.has_p else None,
                                           layout.grow.data_ptr(), layout.grow_host.ctypes.data, len(layout.grow_host),
                                           dx.data_ptr(), _ptr(dps), _ptr(dpe), F, B, N, F, L.dtype_code(x),
                                           L.stream_ptr()), "vmr_map2d_pool_bwd")
        return dx, dps, dpe, None


def map2d_pool(x, ps, pe, layout: Map2dLayout):
    return _Map2dPool.apply(x, ps, pe, layout)


class _Map2dScatter(torch.autograd.Function):
    """dense [B,N,N,W] from compact cells [B,C,W]; off-mask cells = fill[W] (a constant: no gradient)."""

    @staticmethod
    def forward(ctx, cells, fill, layout):
        L.require_gpu(cells)
        B, Cc, W = cells.shape
        assert Cc == layout.C
        cells = cells.contiguous()
        fill = None if fill is None else fill.detach().float().contiguous()
        out = torch.empty(B, layout.N, layout.N, W, device=cells.device, dtype=cells.dtype)
        L.check(L.lib().vmr_map2d_scatter(cells.data_ptr(), layout.cell_of.data_ptr(), _ptr(fill), out.data_ptr(), B,
                                          layout.N, W, Cc, L.dtype_code(cells), L.stream_ptr()), "vmr_map2d_scatter")
        ctx.layout = layout
        return out

    @staticmethod
    def backward(ctx, dout):
        lay = ctx.layout
        return dout[:, lay.ii_t, lay.jj_t, :].contiguous(), None, None


def map2d_scatter(cells, fill, layout: Map2dLayout):
    return _Map2dScatter.apply(cells, fill, layout)
