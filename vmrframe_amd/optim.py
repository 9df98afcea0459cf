"""Fused optimizer step for the SeqPAN train loop (reference main.py:93-97 +
utils/utils.py:87-97): AdamW with the reference's two decay groups (0.01 for
weights, 0 for names containing bias / layer_norm), clip_grad_norm_(max_norm)
and the transformers linear-warmup schedule -- as TWO HIP launches over one flat
fp32 arena (vmr_sumsq + vmr_adamw) instead of ~170 per-tensor updates.

Memory layout: after the first backward, every parameter that received a
gradient is re-pointed (`p.data`, `p.grad`) at views of two contiguous fp32
arenas `flat_p` / `flat_g`; the gradient arena is what the data-parallel
all-reduce moves over xGMI (vmrframe_amd/dp.py).  Parameters that never get a
gradient (the reference's 20 unused tensors, frozen GloVe rows) stay outside
and are never touched, exactly like torch.optim.AdamW skipping `grad is None`.
"""
from __future__ import annotations

from typing import List

import torch

from . import _lib as L

NO_DECAY = ("bias", "layer_norm", "LayerNorm")   # reference utils/utils.py:89


def linear_warmup_lambda(num_warmup_steps: float, num_training_steps: int):
    """transformers.get_linear_schedule_with_warmup's lr multiplier (reference utils/utils.py:95-96)."""
    def f(step: int) -> float:
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return f


class FlatArena:
    """Contiguous fp32 parameter / gradient arenas over the parameters that receive
    gradients (device-agnostic: pure tensor plumbing, also used by the gloo DP tests)."""

    def __init__(self, model: torch.nn.Module, bf16_mirror: bool = False, mirror_dtype=None):
        """mirror_dtype: torch.bfloat16 / torch.float16 = keep a 16-bit copy of every master (and the K-major weight
        copies) in that dtype, rewritten by the AdamW kernel; None = fp32 only.  (bf16_mirror=True: the round-1 spelling
        of mirror_dtype=torch.bfloat16.)"""
        if mirror_dtype is None and bf16_mirror:
            mirror_dtype = torch.bfloat16
        assert mirror_dtype in (None, torch.bfloat16, torch.float16)
        self.mirror_dtype = mirror_dtype
        named = [(n, p) for n, p in model.named_parameters() if p.requires_grad and p.grad is not None]
        assert named, "FlatArena needs one backward pass first (to learn which parameters get gradients)"
        # weights that feed ONE grouped GEMM (q|k|v, ...) are laid out back to back so their
        # compute-dtype mirror is a single [sum N, K] matrix without any per-step concat / cast
        groups = model.weight_groups() if hasattr(model, "weight_groups") else []
        have = dict(named)
        # forward-order stages (model.param_segment): the arena is laid out stage by stage, so a finished stage of
        # the backward pass is ONE contiguous range that data-parallel ranks can all-reduce while earlier stages run
        seg_of = (lambda n: model.param_segment(n)) if hasattr(model, "param_segment") else (lambda n: 0)
        nseg = 1 + max(seg_of(n) for n, _ in named)
        # Within a stage: first the weight MATRICES that are consumed only through their 16-bit mirror / K-major copy (the
        # GEMM operands: the members of the transposed set below), then everything the kernels read in fp32 (biases,
        # LayerNorm and depthwise-conv parameters, small tables, model.fp32_consumed()).  A sharded optimizer
        # (dp.ShardedReducer) reduce-scatters the first region and all-gathers its mirror; the second, small region is
        # all-reduced and updated by every rank.  Region ends are padded to 64 elements, so a region splits evenly
        # (into 16-byte aligned shards) over 2, 4 or 8 ranks.  Without a mirror every parameter is fp32-consumed.
        fp32_names = set(model.fp32_consumed()) if hasattr(model, "fp32_consumed") else set()

        def gemm_matrix(q):
            return q.dim() >= 2 and min(q.shape[0], q.numel() // q.shape[0]) >= 64 and q.numel() % 8 == 0

        def is_mat(names):
            if mirror_dtype is None or any(n in fp32_names for n in names):
                return False
            ps = [have[n] for n in names]
            if len(ps) > 1:      # a grouped projection: one [sum N, K] matrix
                K = ps[0].numel() // ps[0].shape[0]
                return all(q.dim() >= 2 and q.numel() // q.shape[0] == K and q.numel() % 8 == 0 for q in ps) and \
                    min(sum(q.shape[0] for q in ps), K) >= 64
            return gemm_matrix(ps[0])

        plan = []      # (stage, [units of the matrix region], [units of the fp32 region]); unit = list of names kept adjacent
        for sg in range(nseg):
            units = [list(grp) for grp in groups if all(g in have for g in grp) and seg_of(grp[0]) == sg]
            assert all(seg_of(n) == sg for u in units for n in u), "a weight group straddles two stages"
            seen = {n for u in units for n in u}
            units += [[n] for n, _ in named if seg_of(n) == sg and n not in seen]
            plan.append((sg, [u for u in units if is_mat(u)], [u for u in units if not is_mat(u)]))
        dev = named[0][1].device
        total, offs = 0, {}
        self.segment_split = []        # per stage: (lo, mid, hi) = matrix region [lo, mid), fp32 region [mid, hi)
        order = []
        for sg, mats, vecs in plan:
            lo = total
            for region in (mats, vecs):
                for u in region:
                    for n in u:
                        offs[n] = total
                        total += (have[n].numel() + 7) // 8 * 8        # 16-byte aligned slots in the fp32 AND the bf16 arena
                        order.append(n)
                total = (total + 63) // 64 * 64
                if region is mats:
                    mid = total
            self.segment_split.append((lo, mid, total))
        named = [(n, have[n]) for n in order]
        self.flat_p = torch.zeros(total, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.decay = torch.zeros(total, device=dev, dtype=torch.uint8)
        for n, p in named:
            o, k = offs[n], p.numel()
            self.flat_p[o:o + k].copy_(p.data.reshape(-1))
            self.flat_g[o:o + k].copy_(p.grad.reshape(-1))
            p.data = self.flat_p[o:o + k].view(p.shape)
            p.grad = self.flat_g[o:o + k].view(p.shape)
            p._vmr_main_grad = p.grad      # weight-gradient kernels accumulate straight into the arena
            if not any(nd in n for nd in NO_DECAY):
                self.decay[o:o + k] = 1
        self.names: List[str] = [n for n, _ in named]
        self.params = [p for _, p in named]
        self.offsets = offs
        # [lo, hi) of every stage inside the arenas (empty stages: lo == hi)
        self.segment_ranges = [(lo, hi) for lo, _, hi in self.segment_split]
        assert self.segment_ranges[-1][1] == total and all(a[1] == b[0] for a, b in zip(self.segment_ranges[:-1], self.segment_ranges[1:]))
        self.flat_w = None
        self._t_count = 0
        if mirror_dtype is not None:
            # 16-bit copy of every master weight, refreshed by the fused AdamW kernel itself
            self.flat_w = self.flat_p.to(mirror_dtype)
            for n, p in named:
                o, k = offs[n], p.numel()
                p._vmr_w16 = self.flat_w[o:o + k].view(p.shape)
            # K-major (transposed) bf16 copy of every weight MATRIX as the GEMMs use it -- a whole group [sum N, K]
            # for grouped projections, a single [N, K] otherwise -- rewritten after each optimizer step by ONE
            # batched transpose launch (refresh_transposed): the dX products then read weights k-contiguously
            self.flat_wt = torch.zeros_like(self.flat_w)
            items = []
            for sg, mats, vecs in plan:
                for u in mats + vecs:
                    ps = [have[n] for n in u]
                    if len(ps) > 1:
                        K = ps[0].numel() // ps[0].shape[0]
                        if all(q.dim() >= 2 and q.numel() // q.shape[0] == K and q.numel() % 8 == 0 for q in ps):
                            items.append((u[0], ps, sum(q.shape[0] for q in ps), K))
                    elif ps[0].dim() >= 2 and min(ps[0].shape[0], ps[0].numel() // ps[0].shape[0]) >= 64:
                        items.append((u[0], ps, ps[0].shape[0], ps[0].numel() // ps[0].shape[0]))
            self._t_items = (L.TransposeItem * len(items))()
            for it, (n0, ps, rows, cols) in zip(self._t_items, items):
                o = offs[n0]
                it.src, it.dst = self.flat_w[o:].data_ptr(), self.flat_wt[o:].data_ptr()
                it.rows, it.cols = rows, cols
                view = self.flat_wt[o:o + rows * cols].view(cols, rows)
                key = tuple(id(q) for q in ps)
                d = getattr(ps[0], "_vmr_wt_views", None) or {}
                d[key] = view
                ps[0]._vmr_wt_views = d
            self._t_count = len(items)
        for p in self.params:
            p._vmr_arena = self
        self.mark_synced()

    def refresh_transposed(self):
        if getattr(self, "_t_count", 0):
            L.check(L.lib().vmr_transpose_batched(self._t_items, self._t_count, L.stream_ptr()), "vmr_transpose_batched")

    def mark_synced(self):
        """Remember every master's version counter: ops.WeightCache compares against it to notice in-place edits
        (load_state_dict, `with no_grad(): p.copy_(...)`) that the bf16 mirrors have not seen yet."""
        for p in self.params:
            p._vmr_synced_version = p._version

    def sync_mirrors(self):
        """Re-derive the bf16 mirror and the K-major copies from the fp32 masters.  The AdamW kernel keeps them in
        step on its own; this is for changes made to the masters BEHIND the optimizer's back (checkpoint load,
        parameter broadcast, manual `p.data` edits)."""
        if self.flat_w is not None:
            n = self.flat_p.numel()
            L.check(L.lib().vmr_cast(self.flat_p.data_ptr(), L.F32, self.flat_w.data_ptr(), L.dtype_code(self.flat_w), n // 8, 8, 8, 8,
                                     0.0, 0, None, L.stream_ptr()), "vmr_cast")
            self.refresh_transposed()
        self.mark_synced()


class FlatAdamW:
    def __init__(self, model: torch.nn.Module, lr: float, weight_decay: float = 0.01, betas=(0.9, 0.999),
                 eps: float = 1e-8, max_norm: float = 1.0, warmup_steps: float = 0.0, total_steps: int = 0,
                 loss_scale="auto", growth_interval: int = 200, decay_exempt=None):
        """loss_scale: "auto" = a dynamic loss scale (initial 2^16) when the model computes in fp16, none otherwise; a
        number = the initial scale S of a dynamic scaler; None / 0 = off.  With a scale, run the backward pass through
        `opt.backward(loss)` (= (S * loss).backward(), S read on the device): the weight gradients come out scaled by
        S, the AdamW kernel divides by S, and a step whose gradient norm is not finite is SKIPPED on the device (no
        host sync) while vmr_loss_scale_update halves S; `growth_interval` clean steps in a row double it.
        decay_exempt: name fragments of parameters without weight decay (default: the reference's bias / layer_norm
        rule, utils/utils.py:89)."""
        self.model = model
        cd = getattr(model, "compute_dtype", None)
        if loss_scale == "auto":
            loss_scale = 65536.0 if cd == torch.float16 else None
        self.init_scale = float(loss_scale) if loss_scale else 0.0
        self.growth_interval = int(growth_interval)
        self.scale_state = None    # device float[2] = {S, clean-step streak}
        self.decay_exempt = tuple(decay_exempt) if decay_exempt is not None else None
        self.base_lr, self.wd, self.betas, self.eps, self.max_norm = lr, weight_decay, betas, eps, max_norm
        self.sched = linear_warmup_lambda(warmup_steps, total_steps) if total_steps > 0 else (lambda s: 1.0)
        self.t = 0                 # optimizer steps taken (host mirror)
        self.arena = None
        self.m = self.v = self.gnorm_sq = None
        self.warmup_steps, self.total_steps = float(warmup_steps), int(total_steps)
        self.step_t = None   # device-resident step count: the kernel evaluates the lr schedule from it
        self.shard = None    # a dp.ShardedReducer registers itself here: step() then updates this rank's slices only

    # -- arena -----------------------------------------------------------------
    def _build(self):
        cd = getattr(self.model, "compute_dtype", None)
        self.arena = FlatArena(self.model, mirror_dtype=cd if L.is_16bit(cd) else None)
        if self.decay_exempt is not None:
            A = self.arena
            A.decay.zero_()
            for n, p in zip(A.names, A.params):
                if not any(nd in n for nd in self.decay_exempt):
                    A.decay[A.offsets[n]:A.offsets[n] + p.numel()] = 1
        L.require_gpu(self.arena.flat_p)
        self.m = torch.zeros_like(self.arena.flat_p)
        self.v = torch.zeros_like(self.arena.flat_p)
        dev = self.arena.flat_p.device
        self.gnorm_sq = torch.zeros(1, device=dev, dtype=torch.float32)
        self.step_t = torch.full((1,), self.t, device=dev, dtype=torch.int32)
        if self.init_scale > 0 and self.scale_state is None:
            self.scale_state = torch.tensor([self.init_scale, 0.0], device=dev, dtype=torch.float32)
        self.arena.refresh_transposed()
        # a checkpoint loaded into the model from now on lands in the fp32 arena (in-place copies): bring the
        # compute-dtype copies along (reference main.py:26-28 / utils/utils.py:208-215 resume-and-eval flow)
        self.model.register_load_state_dict_post_hook(lambda _m, _keys: self.sync_mirrors())

    def sync_mirrors(self):
        """Call after changing parameter values behind the optimizer's back (dp.broadcast_parameters, manual
        `p.data` edits); load_state_dict and in-place `p.copy_()` edits are noticed without it."""
        if self.arena is not None:
            self.arena.sync_mirrors()
        if hasattr(self.model, "_cache"):
            self.model._cache.clear()

    @property
    def grad_arena(self):
        return None if self.arena is None else self.arena.flat_g

    @property
    def names(self):
        return self.arena.names

    @property
    def offsets(self):
        return self.arena.offsets

    def zero_grad(self):
        cache = getattr(self.model, "_cache", None)
        if cache is not None and hasattr(cache, "state"):
            cache.state.reset()             # a backward pass that died must not leak partials into this one
        if self.arena is None:
            for p in self.model.parameters():
                p.grad = None
        else:
            self.arena.flat_g.zero_()       # p.grad stay views: autograd accumulates in place

    def lr(self) -> float:
        return self.base_lr * self.sched(self.t)

    def step(self):
        if self.arena is None:
            self._build()
        sh = self.shard
        if sh is not None and (sh.world > 1 or getattr(sh, "force", False)):
            # sharded optimizer (dp.ShardedReducer): three device parts around two exchanges
            self.step_norm()
            sh.sum_scalar(self.gnorm_sq)
            self.step_update()
            sh.gather(self.arena.flat_w if self.arena.flat_w is not None else self.arena.flat_p)
            self.step_finish()
            return
        A = self.arena
        lib, st = L.lib(), L.stream_ptr()
        self.gnorm_sq.zero_()
        n = A.flat_p.numel()
        L.check(lib.vmr_sumsq(A.flat_g.data_ptr(), self.gnorm_sq.data_ptr(), n, st), "vmr_sumsq")
        self._adamw(0, n)
        A.refresh_transposed()       # K-major weight copies follow the 16-bit mirror the kernel just rewrote
        self._advance()

    def _adamw(self, lo: int, hi: int):
        """vmr_adamw over [lo, hi) of the arenas (16-byte aligned bounds)."""
        if hi <= lo:
            return
        A, sc = self.arena, self.scale_state
        assert lo % 8 == 0
        L.check(L.lib().vmr_adamw(A.flat_p[lo:].data_ptr(), A.flat_g[lo:].data_ptr(), self.m[lo:].data_ptr(), self.v[lo:].data_ptr(),
                                  A.decay[lo:].data_ptr(), None if A.flat_w is None else A.flat_w[lo:].data_ptr(),
                                  L.BF16 if A.flat_w is None else L.dtype_code(A.flat_w),
                                  self.gnorm_sq.data_ptr(), self.max_norm, self.base_lr,
                                  self.betas[0], self.betas[1], self.eps, self.wd, 0, self.step_t.data_ptr(),
                                  self.warmup_steps, float(self.total_steps), None if sc is None else sc.data_ptr(), hi - lo,
                                  L.stream_ptr()), "vmr_adamw")

    def _advance(self):
        sc = self.scale_state
        if sc is None:
            self.step_t += 1         # scheduler.step() of the reference loop
        else:                        # the same, unless the step overflowed (then S halves and the count holds)
            L.check(L.lib().vmr_loss_scale_update(sc.data_ptr(), self.gnorm_sq.data_ptr(), self.step_t.data_ptr(),
                                                  self.growth_interval, 1.0, L.stream_ptr()), "vmr_loss_scale_update")
        self.t += 1
        if hasattr(self.model, "_cache"):   # the masters changed under the compute-dtype weight cache
            self.model._cache.clear()

    # -- the three device parts of a SHARDED step (dp.ShardedReducer; each is capturable on its own) ----------------
    def step_norm(self):
        """Part 1: the sum of squares of THIS rank's slices of the averaged gradient (the matrix regions are reduce-
        scattered: nobody holds them whole); the ranks' partial sums are added by one scalar all-reduce."""
        if self.arena is None:
            self._build()
        lib, st = L.lib(), L.stream_ptr()
        self.gnorm_sq.zero_()
        g = self.arena.flat_g
        for lo, hi in self.shard.my_slices():
            if hi > lo:
                L.check(lib.vmr_sumsq(g[lo:].data_ptr(), self.gnorm_sq.data_ptr(), hi - lo, st), "vmr_sumsq")

    def step_update(self):
        """Part 2 (gnorm_sq now holds the matrix regions' total): add the all-reduced fp32 regions' squares -- every rank
        holds those whole -- then AdamW on this rank's slices and on the fp32 regions."""
        lib, st = L.lib(), L.stream_ptr()
        g = self.arena.flat_g
        for lo, hi in self.shard.fp32_regions():
            if hi > lo:
                L.check(lib.vmr_sumsq(g[lo:].data_ptr(), self.gnorm_sq.data_ptr(), hi - lo, st), "vmr_sumsq")
        for lo, hi in self.shard.my_slices() + self.shard.fp32_regions():
            self._adamw(lo, hi)
        self._advance()

    def step_finish(self):
        """Part 3 (the mirrors of the foreign slices have been gathered): the K-major weight copies."""
        self.arena.refresh_transposed()

    def grad_norm(self) -> float:
        """the (unscaled) global gradient norm of the last step; inf / nan if that step overflowed and was skipped"""
        g = float(self.gnorm_sq.sqrt().item())
        return g if self.scale_state is None else g / self._last_scale()

    def _last_scale(self) -> float:
        return float(self.scale_state[0].item())

    def loss_scale(self) -> float:
        """current loss scale S (1.0 when scaling is off); a device read"""
        return 1.0 if self.scale_state is None else self._last_scale()

    def scaled(self, loss: torch.Tensor) -> torch.Tensor:
        """S * loss with S read on the device (graph-capturable); the loss itself when scaling is off"""
        if self.init_scale <= 0:
            return loss
        if self.scale_state is None:        # first pass, before the arena exists
            self.scale_state = torch.tensor([self.init_scale, 0.0], device=loss.device, dtype=torch.float32)
        return loss * self.scale_state[0]

    def backward(self, loss: torch.Tensor, run=None):
        """loss.backward() with the loss scale applied.  `run`: a callable that runs the backward pass for the given
        (scaled) loss -- e.g. model.segmented_backward -- default .backward()."""
        loss = self.scaled(loss)
        return loss.backward() if run is None else run(loss)
