"""BAN's sequence encoders on the HIP library (second slice): the bidirectional LSTM of
`QueryEncoder` / `VisualEncoder` (reference models/BANlib/model.py:8-86).

The reference runs `nn.LSTM(input, hidden, 1, batch_first=True, bidirectional=True)` on a sequence packed by length
(`pack_padded_sequence(..., enforce_sorted=False)`), pads the output back and mean-pools the valid steps.  Here:

  * the input projections of all steps are two `vmr_gemm` products (direction 1 on the per-sample REVERSED sequence,
    `vmr_lstm_reverse_rows`), both biases folded into their epilogue;
  * the recurrence runs in STEP order for both directions at once: step s = time s for direction 0 and time
    len_b - 1 - s for direction 1, both active while s < len_b -- which is what packing computes (a sample's reverse
    pass starts at its own last valid step from a zero state) with one mask for both directions.  Per step: one
    batched product h . W_hh^T (both directions, `Z1 = 2`) and one `vmr_lstm_cell_fwd` launch (gates, state, the scatter
    of h into the time-ordered [B, T, 2H] output);
  * backward: per step `vmr_lstm_cell_bwd` + the batched product dg_s . W_hh; after the loop ONE product per weight over
    all steps (dW_hh = dg^T . h_prev, dW_ih = dg^T . x, K = B*T), the bias gradients as column sums, dx = dg . W_ih with
    direction 1 mapped back through the same reversal.
  * no per-sample host loop: the mean over valid steps is sum_t / len (the output is zero past len).

State-dict keys are the reference's (`biLSTM.weight_ih_l0`, `..._reverse`, ...), so a BAN checkpoint's encoder
weights load unchanged.  The recurrence is launch-serial by construction (one fused launch per step, bf16, H = 256 / 512;
product + cell launches otherwise); a persistent recurrence kernel is the next step.  QueryEncoder's
embedding front (`F.embedding` over [pad | unk | glove]) is ops.embedding.  The rest of BAN: ban_trunk.py, ban_map.py,
ban_sampler.py, ban_head.py, assembled in ban.py.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import ops


FUSED_STEP = os.environ.get("VMR_LSTM_FUSED", "1") != "0"      # A/B: one launch per step (bf16, H = 256 / 512) vs product + cell


def _reverse_rows(x: torch.Tensor, lens: torch.Tensor) -> torch.Tensor:
    B, T, D = x.shape
    out = torch.empty_like(x)
    L.check(L.lib().vmr_lstm_reverse_rows(x.data_ptr(), lens.data_ptr(), out.data_ptr(), B, T, D, L.dtype_code(x),
                                          L.stream_ptr()), "vmr_lstm_reverse_rows")
    return out


SEQ_KERNEL = os.environ.get("VMR_LSTM_SEQ", "1") != "0"     # one persistent launch per layer and pass (csrc/lstm.hip)
_SYNC_LOG = []       # the sync words of the most recent sequence launches (tests read word 8: a workgroup gave up waiting)


_ERR_WORDS = {}      # device -> the shared, never-reset sync words of the sentinel-exchange launches (only word 8, the error flag, is ever written)


def _sync_words(dev, counters: bool):
    """The 16 sync words of one sequence launch: fresh zeros when the launch counts arrivals in them; with the sentinel
    exchange only the error word is used (written on a poll timeout, sticky), so every launch shares one buffer -- no
    fill launch per recurrence."""
    if counters:
        return torch.zeros(16, device=dev, dtype=torch.int32)
    t = _ERR_WORDS.get(dev)
    if t is None:
        if torch.cuda.is_current_stream_capturing():      # (a graph's private pool is no home for a process-wide buffer)
            return torch.zeros(16, device=dev, dtype=torch.int32)
        t = _ERR_WORDS[dev] = torch.zeros(16, device=dev, dtype=torch.int32)
    return t


def _note_sync(sync):
    _SYNC_LOG.append(sync)
    del _SYNC_LOG[:-64]


def seq_kernel_gave_up() -> bool:
    """True if any of the recent one-launch recurrences raised its error word (a device read: tests / debugging only)."""
    return any(int(t[8].item()) != 0 for t in _SYNC_LOG)


def _recur_fwd(gx, w_hh, lens, act, cs, hp, y, B, T, H, Z, dt, dc) -> str:
    """The recurrence over all T steps for Z directions (gx = the x-part of every step's gates, biases included): fills
    act / cs / hp (saved for the backward pass) and scatters h into y.  Returns the form it ran in -- "seq": ONE persistent
    launch; "fused": product + gates + state in one launch per step; "plain": a product and a cell launch per step."""
    lib, dev = L.lib(), gx.device
    fused = bool(FUSED_STEP and lib.vmr_lstm_step_supported(H, dc))
    if fused and SEQ_KERNEL and lib.vmr_lstm_seq_supported(B, H, Z, dc):
        nbytes = C.c_int64(0)
        L.check(lib.vmr_lstm_seq_hist_bytes(T, H, Z, C.byref(nbytes)), "vmr_lstm_seq_hist_bytes")
        hist = torch.empty(nbytes.value // 2, device=dev, dtype=torch.int16)
        if lib.vmr_lstm_seq_sentinel():      # the exchange reads readiness off the history itself: "not yet written" everywhere
            hist.fill_(0x7FFF)
        sync = _sync_words(dev, counters=not lib.vmr_lstm_seq_sentinel())
        L.check(lib.vmr_lstm_seq_fwd(gx.data_ptr(), w_hh.data_ptr(), lens.data_ptr(), act.data_ptr(), cs.data_ptr(),
                                     hp.data_ptr(), y.data_ptr(), hist.data_ptr(), sync.data_ptr(), B, T, H, Z, dc,
                                     L.stream_ptr()), "vmr_lstm_seq_fwd")
        _note_sync(sync)
        return "seq"
    c = torch.zeros(Z, B, H, device=dev)
    if fused:                    # h ping-pongs between two buffers
        hb = torch.zeros(2, Z, B, H, device=dev, dtype=dt)
        for s in range(T):
            L.check(lib.vmr_lstm_step_fwd(gx.data_ptr(), hb[s & 1].data_ptr(), w_hh.data_ptr(), lens.data_ptr(),
                                          c.data_ptr(), hb[(s + 1) & 1].data_ptr(), act.data_ptr(), cs.data_ptr(),
                                          hp.data_ptr(), y.data_ptr(), B, T, H, s, Z, dc, L.stream_ptr()),
                    "vmr_lstm_step_fwd")
        return "fused"
    hs = torch.zeros(Z, B, H, device=dev, dtype=dt)
    gh = torch.zeros(Z, B, 4 * H, device=dev)
    f32out = L.EPI_OUT_F32 if dc != L.F32 else 0
    for s in range(T):
        if s > 0:                                                 # gh[z] = hs[z] . w_hh[z]^T, every direction
            ops.gemm(hs, w_hh, gh, B, 4 * H, H, 0, 0, H, H, 4 * H, dtype=dc, flags=f32out, Z1=Z,
                     sA=(B * H, 0), sB=(4 * H * H, 0), sC=(B * 4 * H, 0))
        L.check(lib.vmr_lstm_cell_fwd(gx.data_ptr(), gh.data_ptr(), lens.data_ptr(), c.data_ptr(), hs.data_ptr(),
                                      act.data_ptr(), cs.data_ptr(), hp.data_ptr(), y.data_ptr(), B, T, H, s, Z, dc,
                                      L.stream_ptr()), "vmr_lstm_cell_fwd")
    return "plain"


def _recur_bwd(dy, act, cs, lens, w_hh, mode, B, T, H, Z, dt, dc) -> torch.Tensor:
    """dg [Z, B, T, 4H] (gradient of every step's pre-activation gates) from dy [Z/2, B, T, 2H], in the form the forward ran in."""
    lib, dev = L.lib(), dy.device
    dg = torch.empty(Z, B, T, 4 * H, device=dev, dtype=dt)
    if mode == "seq":
        if lib.vmr_lstm_seq_bwd_sentinel():      # dg is the exchange buffer and carries its own readiness
            dg.view(torch.int16).fill_(0x7FFF)
        whht = w_hh.transpose(1, 2).contiguous()                      # [Z, H, 4H]: the K-contiguous operand of dg . W_hh
        sync = _sync_words(dev, counters=not lib.vmr_lstm_seq_bwd_sentinel())
        L.check(lib.vmr_lstm_seq_bwd(dy.data_ptr(), act.data_ptr(), cs.data_ptr(), lens.data_ptr(), whht.data_ptr(),
                                     dg.data_ptr(), sync.data_ptr(), B, T, H, Z, dc, L.stream_ptr()), "vmr_lstm_seq_bwd")
        _note_sync(sync)
        return dg
    dcell = torch.zeros(Z, B, H, device=dev)
    if mode == "fused":
        whht = w_hh.transpose(1, 2).contiguous()
        for s in range(T - 1, -1, -1):
            L.check(lib.vmr_lstm_step_bwd(dy.data_ptr(), act.data_ptr(), cs.data_ptr(), lens.data_ptr(), whht.data_ptr(),
                                          dcell.data_ptr(), dg.data_ptr(), B, T, H, s, Z, dc, L.stream_ptr()),
                    "vmr_lstm_step_bwd")
        return dg
    dh = torch.zeros(Z, B, H, device=dev)
    f32out = L.EPI_OUT_F32 if dc != L.F32 else 0
    for s in range(T - 1, -1, -1):
        L.check(lib.vmr_lstm_cell_bwd(dy.data_ptr(), act.data_ptr(), cs.data_ptr(), lens.data_ptr(), dh.data_ptr(),
                                      dcell.data_ptr(), dg.data_ptr(), B, T, H, s, Z, dc, L.stream_ptr()),
                "vmr_lstm_cell_bwd")
        if s > 0:  # dh[z] = dg[z][:, s, :] . w_hh[z]   (A rows strided by T*4H; W_hh is the [K][N] operand)
            ops.gemm(dg[:, :, s], w_hh, dh, B, H, 4 * H, 0, 1, T * 4 * H, H, H, dtype=dc, flags=f32out, Z1=Z,
                     sA=(B * T * 4 * H, 0), sB=(4 * H * H, 0), sC=(B * H, 0))
    return dg


class _BiLSTM(torch.autograd.Function):
    """y [K, B, T, 2H] = K independent bi-LSTMs advanced together (x [K, B, T, I], lens int32 [B] shared);
    w_ih [2K, 4H, I], w_hh [2K, 4H, H], bias [2K, 4H] (= b_ih + b_hh; rows 2k / 2k + 1 = forward / reverse direction of LSTM k),
    all in x's dtype (fp32 or bf16).  Gradients for x, w_ih, w_hh, bias.  K = 1 is the plain encoder; K = 2 is
    TemporalDifference's pair (same shapes, own weights): one launch per step serves both."""

    @staticmethod
    def forward(ctx, x, lens, w_ih, w_hh, bias):
        L.require_gpu(x, lens, w_ih, w_hh, bias)
        assert lens.dtype == torch.int32 and x.is_contiguous() and w_ih.is_contiguous() and w_hh.is_contiguous()
        K, B, T, I = x.shape
        Z, H = 2 * K, w_hh.shape[2]
        assert w_ih.shape[0] == Z and w_hh.shape[0] == Z
        dt, dc = x.dtype, L.dtype_code(x)
        lib, dev = L.lib(), x.device
        xr = _reverse_rows(x.view(K * B, T, I), lens.repeat(K)).view(K, B, T, I)
        xs = torch.stack((x, xr), dim=1).view(Z, B, T, I)                  # [Z, B, T, I] by step (z = 2k + direction)
        gx = torch.empty(Z, B, T, 4 * H, device=dev, dtype=dt)
        bias32 = bias.float().contiguous()
        for z in range(Z):                                                # x-part of every step, biases in the epilogue
            ops.mm(xs[z].view(B * T, I), w_ih[z], 0, 0, out=gx[z].view(B * T, 4 * H), bias=bias32[z], flags=L.EPI_BIAS)
        act = torch.empty(Z, B, T, 4 * H, device=dev, dtype=dt)
        cs = torch.empty(Z, B, T, H, device=dev)
        hp = torch.empty(Z, B, T, H, device=dev, dtype=dt)
        y = torch.zeros(K, B, T, 2 * H, device=dev, dtype=dt)
        ctx.mode = _recur_fwd(gx, w_hh, lens, act, cs, hp, y, B, T, H, Z, dt, dc)
        ctx.save_for_backward(xs, lens, w_ih, w_hh, act, cs, hp)
        ctx.mark_non_differentiable(lens)
        return y

    @staticmethod
    def backward(ctx, dy):
        xs, lens, w_ih, w_hh, act, cs, hp = ctx.saved_tensors
        Z, B, T, I = xs.shape
        K, H = Z // 2, w_hh.shape[2]
        dt, dc = xs.dtype, L.dtype_code(xs)
        lib, dev = L.lib(), xs.device
        dg = _recur_bwd(dy.contiguous(), act, cs, lens, w_hh, ctx.mode, B, T, H, Z, dt, dc)
        dg2 = dg.view(Z, B * T, 4 * H)
        dw_hh = torch.empty(Z, 4 * H, H, device=dev)
        dw_ih = torch.empty(Z, 4 * H, I, device=dev)
        dxs = torch.empty(Z, B, T, I, device=dev, dtype=dt)
        for z in range(Z):   # one product per weight over all steps (K = B*T); dx of each direction's step sequence
            _wgrad(dg2[z], hp[z].view(B * T, H), dw_hh[z])
            _wgrad(dg2[z], xs[z].view(B * T, I), dw_ih[z])
            ops.mm(dg2[z], w_ih[z], 0, 1, out=dxs[z].view(B * T, I))
        # bias gradient = column sums of dg over all (sample, step) rows: one column-sum launch per direction on the 16-bit
        # dg itself (as `dg2.float().sum(1)` it was an fp32 copy of dg + a reduction: 0.5 ms per BAN step)
        dbias = torch.zeros(Z, 4 * H, device=dev, dtype=torch.float32)
        for z in range(Z):
            L.check(lib.vmr_relu_bwd_bias(0, dg2[z].data_ptr(), None, None, dbias[z].data_ptr(), B * T, 4 * H, 4 * H, 1.0, dc, 0.0, 0,
                                          None, None, 1.0, L.stream_ptr()), "vmr_relu_bwd_bias")
        dxs = dxs.view(K, 2, B, T, I)                                    # (the reversal is its own inverse; zero past len)
        dx = dxs[:, 0] + _reverse_rows(dxs[:, 1].reshape(K * B, T, I), lens.repeat(K)).view(K, B, T, I)
        return dx, None, dw_ih.to(w_ih.dtype), dw_hh.to(w_hh.dtype), dbias.to(dt)


def bilstm(x, lens, w_ih, w_hh, bias):
    """One bi-LSTM: x [B, T, I] -> [B, T, 2H].  lens is clamped to [0, T]: the kernels index time by it."""
    return _BiLSTM.apply(x.unsqueeze(0), lens.clamp(min=0, max=x.shape[1]), w_ih, w_hh, bias)[0]


def bilstm_multi(x, lens, w_ih, w_hh, bias):
    """K independent bi-LSTMs of the same shape advanced by the same launches: x [K, B, T, I] -> [K, B, T, 2H];
    w_ih [2K, 4H, I], w_hh [2K, 4H, H], bias [2K, 4H] (LSTM k: rows 2k forward, 2k + 1 reverse)."""
    return _BiLSTM.apply(x.contiguous(), lens.clamp(min=0, max=x.shape[2]), w_ih, w_hh, bias)


PARAM_DIRECT = os.environ.get("VMR_LSTM_PARAM_DIRECT", "1") != "0"   # A/B: the layer reads master parameters / arena mirrors itself


def _w16(p: torch.Tensor, dt: torch.dtype, kpad: int = 0) -> torch.Tensor:
    """A parameter matrix in the compute dtype without a torch op where possible: the 16-bit mirror the AdamW kernel keeps
    in the flat arena (optim.FlatArena), the fp32 master itself for the fp32 path, one vmr_cast launch otherwise (also
    when the K columns need zero padding to 16-byte rows)."""
    K = p.shape[1]
    Kp = max(K, kpad)
    m = getattr(p, "_vmr_w16", None)
    if m is not None and m.dtype == dt and Kp == K:
        if p._version != getattr(p, "_vmr_synced_version", p._version):
            p._vmr_arena.sync_mirrors()          # edited in place behind the optimizer's back (load_state_dict)
        return m
    if dt == torch.float32 and Kp == K:
        return p.detach()
    out = torch.empty(p.shape[0], Kp, device=p.device, dtype=dt)
    src = p.detach()
    L.check(L.lib().vmr_cast(src.data_ptr(), L.F32, out.data_ptr(), L.dtype_code(out), p.shape[0], K, K, Kp, 0.0, 0, None,
                             L.stream_ptr()), "vmr_cast")
    return out


class _BiLSTMLayer(torch.autograd.Function):
    """One bi-LSTM layer straight on its eight master parameters (nn.LSTM's weight_ih / weight_hh / bias_ih / bias_hh of
    the forward and the reverse direction): y [B, T, 2H] from x [B, T, I] in the compute dtype.  Same kernels as _BiLSTM;
    what differs is the glue around them -- no stacked / cast weight copies per step (arena mirrors), both bias vectors
    added by the product's epilogue, and in the backward pass weight gradients reduced straight into the flat gradient
    arena with the bias gradients as column sums riding on the same products (no autograd accumulation, no casts)."""

    @staticmethod
    def forward(ctx, x, lens, *params):
        L.require_gpu(x, lens)
        assert lens.dtype == torch.int32 and x.is_contiguous() and len(params) == 8
        B, T, I = x.shape
        H = params[1].shape[1]
        Z = 2
        dt, dc = x.dtype, L.dtype_code(x)
        lib, dev = L.lib(), x.device
        w_ih = [_w16(params[4 * z], dt, kpad=I) for z in range(Z)]          # (x may carry zero columns up to 16-byte rows)
        w_hh = torch.stack([_w16(params[4 * z + 1], dt) for z in range(Z)])
        xs = (x, _reverse_rows(x, lens))
        gx = torch.empty(Z, B, T, 4 * H, device=dev, dtype=dt)
        for z in range(Z):                                                # x-part of every step, b_ih + b_hh in the epilogue
            ops.mm(xs[z].view(B * T, I), w_ih[z], 0, 0, out=gx[z].view(B * T, 4 * H), bias=params[4 * z + 2].detach(),
                   bias2=params[4 * z + 3].detach(), flags=L.EPI_BIAS)
        act = torch.empty(Z, B, T, 4 * H, device=dev, dtype=dt)
        cs = torch.empty(Z, B, T, H, device=dev)
        hp = torch.empty(Z, B, T, H, device=dev, dtype=dt)
        y = torch.zeros(1, B, T, 2 * H, device=dev, dtype=dt)
        ctx.mode = _recur_fwd(gx, w_hh, lens, act, cs, hp, y, B, T, H, Z, dt, dc)
        ctx.save_for_backward(xs[0], xs[1], lens, w_hh, act, cs, hp, *w_ih)
        ctx.params = params
        ctx.mark_non_differentiable(lens)
        return y[0]

    @staticmethod
    def backward(ctx, dy):
        x0, x1, lens, w_hh, act, cs, hp, *w_ih = ctx.saved_tensors
        params = ctx.params
        xs = (x0, x1)
        B, T, I = x0.shape
        Z, H = 2, w_hh.shape[2]
        dt, dc = x0.dtype, L.dtype_code(x0)
        lib, dev = L.lib(), x0.device
        dg = _recur_bwd(dy.contiguous().unsqueeze(0), act, cs, lens, w_hh, ctx.mode, B, T, H, Z, dt, dc)
        dg2 = dg.view(Z, B * T, 4 * H)
        dxs = torch.empty(Z, B, T, I, device=dev, dtype=dt)
        grads = [None] * 8
        for z in range(Z):
            pih, phh, bih, bhh = params[4 * z:4 * z + 4]
            slots = [ops.main_grad(q) for q in (pih, phh, bih, bhh)]
            direct = all(s_ is not None for s_ in slots) and L.is_16bit(dt)
            if direct:        # split-K slabs, ONE reduction each straight into the arena; bias gradients ride on the products
                _wgrad_into(dg2[z], hp[z].view(B * T, H), slots[1], H, slots[3])
                _wgrad_into(dg2[z], xs[z].view(B * T, I), slots[0], pih.shape[1], slots[2])
            else:
                dwh = torch.empty(4 * H, H, device=dev)
                dwi = torch.empty(4 * H, I, device=dev)
                _wgrad(dg2[z], hp[z].view(B * T, H), dwh)
                _wgrad(dg2[z], xs[z].view(B * T, I), dwi)
                db = torch.zeros(4 * H, device=dev)
                L.check(lib.vmr_relu_bwd_bias(0, dg2[z].data_ptr(), None, None, db.data_ptr(), B * T, 4 * H, 4 * H, 1.0, dc, 0.0, 0,
                                              None, None, 1.0, L.stream_ptr()), "vmr_relu_bwd_bias")
                grads[4 * z:4 * z + 4] = [dwi[:, :pih.shape[1]], dwh, db, db]
            ops.mm(dg2[z], w_ih[z], 0, 1, out=dxs[z].view(B * T, I))
        dx = dxs[0] + _reverse_rows(dxs[1], lens)                          # (the reversal is its own inverse; zero past len)
        return (dx, None, *grads)


def _wgrad(a2, b2, out):
    """out [4H, N] fp32 = a2^T . b2 over K = B*T: few output tiles, long K -> split-K slabs + one reduce launch (16-bit);
    the plain product otherwise (fp32, or many tiles)."""
    r = ops.mm_few_tiles(a2, b2, 1, 1)
    if r is None:
        ops.mm(a2, b2, 1, 1, out=out, out_f32=True)
    else:
        out.copy_(r)


def _wgrad_into(a2, b2, slot, kcols, bslot):
    """slot [4H, kcols] (fp32, gradient arena) += a2^T . b2, bslot [4H] += column sums of a2: K = B*T split into fp32
    slabs over the idle CUs + one vmr_splitk_reduce straight into the arena (b2 may carry zero-padded columns: only the
    first kcols of every slab row are reduced); short K: one accumulating product."""
    K, M = a2.shape
    N = b2.shape[1]
    dc = L.dtype_code(a2)
    if K >= 512 and N % 4 == 0:
        tiles = ops._cdiv(M, 128) * ops._cdiv(N, 128)
        sk = 2 if tiles >= 128 else max(2, min(8, K // 256 if tiles >= 32 else K // 128, ops._cdiv(320, tiles)))
        ws = torch.empty(sk, M, N, device=a2.device, dtype=torch.float32)
        ops.gemm(a2, b2, ws, M, N, K, 1, 1, a2.stride(0), b2.stride(0), N, dtype=dc, flags=L.EPI_SLAB, splitk=sk, a_colsum=bslot)
        padded = kcols != N
        L.check(L.lib().vmr_splitk_reduce(ws.data_ptr(), slot.data_ptr(), sk, M * N, N if padded else 0, kcols if padded else 0,
                                          L.stream_ptr()), "vmr_splitk_reduce")
    elif kcols == N:
        ops.gemm(a2, b2, slot, M, N, K, 1, 1, a2.stride(0), b2.stride(0), N, dtype=dc, flags=L.EPI_ACCUM, a_colsum=bslot)
    else:
        slot.view(M, kcols).add_(ops.mm(a2, b2, 1, 1, out_f32=True, a_colsum=bslot)[:, :kcols])


def lstm_layer(m: nn.LSTM, l: int, h: torch.Tensor, lens: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    """Layer l of the bidirectional nn.LSTM parameter holder m on h [B, T, I] (compute dtype): [B, T, 2H].  lens int32 [B]
    must lie in [0, T] (the kernels index time by it): the callers clamp once per stack, not per layer."""
    names = [f"{k}_l{l}{sfx}" for sfx in ("", "_reverse") for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    params = [getattr(m, n) for n in names]
    pad = (-h.shape[2]) % 8                   # 16-byte rows: zero columns on both sides of the first product
    if PARAM_DIRECT:
        if pad:
            h = torch.nn.functional.pad(h, (0, pad))
        return _BiLSTMLayer.apply(h.contiguous(), lens, *params)
    w_ih = torch.stack((params[0], params[4]))
    w_hh = torch.stack((params[1], params[5]))
    bias = torch.stack((params[2] + params[3], params[6] + params[7]))
    if pad:
        h = torch.nn.functional.pad(h, (0, pad))
        w_ih = torch.nn.functional.pad(w_ih, (0, pad))
    return bilstm(h.contiguous(), lens, w_ih.to(dt).contiguous(), w_hh.to(dt).contiguous(), bias.to(dt).contiguous())


class _EncoderBase(nn.Module):
    """Parameter holder with nn.LSTM's key names under `biLSTM.` + the HIP forward (stacked layers: layer l + 1 reads layer
    l's [B, T, 2H] output with the same lengths -- what nn.LSTM does on a packed sequence; inter-layer dropout is 0.0 in
    the reference)."""

    def __init__(self, input_dim: int, hidden_dim: int, num_layers: int = 1, compute_dtype=torch.float32):
        super().__init__()
        assert hidden_dim % 8 == 0, "16-byte rows"
        self.biLSTM = nn.LSTM(input_dim, hidden_dim, num_layers, dropout=0.0, batch_first=True, bidirectional=True)
        self.hidden_dim, self.num_layers, self.compute_dtype = hidden_dim, num_layers, compute_dtype

    def _layers(self, x: torch.Tensor, lens: torch.Tensor):
        """x [B, T, I] (any float dtype), lens int32 [B] on x's device -> (mean over valid steps [B, 2H], output [B, T, 2H])"""
        m, dt = self.biLSTM, self.compute_dtype
        h = x.to(dt)
        lens = lens.clamp(min=0, max=x.shape[1])
        for l in range(self.num_layers):
            h = lstm_layer(m, l, h, lens, dt)
        vec = h.float().sum(1) / lens.clamp(min=1).unsqueeze(1).float()     # mean over the valid steps (zero past len)
        return vec.to(h.dtype), h


class VisualEncoder(_EncoderBase):
    """reference models/BANlib/model.py:60-86: forward(visual_data [B,T,I], visual_length [B], max_seq_len) ->
    (v_vector [B, 2H], output [B, max_seq_len, 2H])."""

    def __init__(self, input_dim=500, hidden_dim=512, num_layers=1, bidirection=True, compute_dtype=torch.float32):
        assert bidirection, "the reference instantiates bidirectional encoders only"
        super().__init__(input_dim, hidden_dim, num_layers, compute_dtype)

    def forward(self, visual_data, visual_length, max_seq_len):
        x = visual_data
        if x.shape[1] != max_seq_len:                                      # (pad_packed_sequence(total_length=...))
            x = x[:, :max_seq_len] if x.shape[1] > max_seq_len else torch.nn.functional.pad(x, (0, 0, 0, max_seq_len - x.shape[1]))
        lens = visual_length.to(device=x.device, dtype=torch.int32).contiguous()
        return self._layers(x, lens)


class QueryEncoder(_EncoderBase):
    """reference models/BANlib/model.py:8-57 with pre-trained vectors: the table is [pad_vec | unk_vec | glove_vec]
    (pad and glove frozen, unk trainable), forward(query_tokens [B,L], query_length [B]) -> (q_vector, output
    [B, max(query_length), 2H])."""

    def __init__(self, vocab_size, hidden_dim=512, embed_dim=300, num_layers=1, bidirection=True, pre_train_weights=None,
                 compute_dtype=torch.float32):
        assert bidirection and pre_train_weights is not None
        super().__init__(embed_dim, hidden_dim, num_layers, compute_dtype)
        self.embed_dim = embed_dim
        self.embedding = nn.Embedding(vocab_size, embed_dim, padding_idx=0)      # (held, unused: as in the reference)
        self.embedding.weight.requires_grad = False
        w = torch.as_tensor(pre_train_weights, dtype=torch.float32)
        self.pad_vec = nn.Parameter(torch.zeros(1, embed_dim), requires_grad=False)
        unk = torch.empty(1, embed_dim)
        nn.init.xavier_uniform_(unk)
        self.unk_vec = nn.Parameter(unk, requires_grad=True)
        self.glove_vec = nn.Parameter(w.clone(), requires_grad=False)

    def forward(self, query_tokens, query_length, max_len=None):
        """max_len (not in the reference): the longest query of the batch when the caller knows it -- spares the
        device-to-host read of `query_length.max()`, which is illegal while a hipGraph is being captured."""
        table = torch.cat([self.pad_vec, self.unk_vec, self.glove_vec], dim=0)
        emb = ops.embedding(query_tokens, table, padding_idx=0)                # [B, L, E] fp32
        Lmax = int(query_length.max()) if max_len is None else int(max_len)   # pad_packed_sequence trims to the longest
        lens = query_length.to(device=emb.device, dtype=torch.int32).contiguous()
        return self._layers(emb[:, :Lmax], lens)
