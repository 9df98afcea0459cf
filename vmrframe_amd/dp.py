"""Batch data parallelism for the SeqPAN path: one process per GPU, each rank
trains on its own shard of clips and gradients are averaged with RCCL
all-reduce over xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

Replaces the reference's single-process nn.DataParallel (reference main.py:22-24),
which cannot work for this model.  Semantics are the
reference's per-replica forward at the local batch (the batch-axis attention of
layers.py:567-574 sees the LOCAL batch) with DDP-style gradient averaging.

Overlap.  Gradients live in one flat fp32 arena laid out stage by stage in forward
order (optim.FlatArena.segment_ranges; stages = SeqPAN.SEGMENT_PREFIXES).  The
backward pass is cut at the stage boundaries (SeqPAN.segmented_backward): as soon
as a stage's piece of the pass has finished -- predictor first, embeddings last --
its contiguous arena range goes to RCCL as ONE large message on RCCL's own stream
while the next piece computes.  Only the first stage's range (text / video
projections + shared encoder, 8 % of the arena) has nothing left to hide behind.
Weight gradients are written straight into the arena by the dW kernels, so there
are no per-parameter autograd hooks to hang the launches on; the cuts are what
makes "this range is final" known on the host (and they are ordinary graph
boundaries for the hipGraph trainer).  Point-to-point xGMI favours few, large
messages: 5 ranges of 20-67 MB instead of ~170 tensors.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None):
    import os
    if dist.is_initialized():
        return
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", 0)),
                            world_size=int(os.environ.get("WORLD_SIZE", 1)))


def shard_batch(batch: Dict[str, torch.Tensor], rank: int, world: int) -> Dict[str, torch.Tensor]:
    """Give rank r the clips r, r+world, ... (whole clips only; no tensor is split inside a clip)."""
    return {k: v[rank::world] for k, v in batch.items()}


def broadcast_parameters(model: torch.nn.Module, src: int = 0, optimizer=None):
    """Every rank starts from rank `src`'s weights.  With a flat arena (optimizer given and built) that is ONE
    broadcast of the fp32 arena plus the few parameters outside it, followed by a refresh of the compute-dtype
    mirrors; before the arena exists, parameters are coalesced into one flat buffer per dtype."""
    import os
    if not dist.is_initialized() or (dist.get_world_size() == 1 and os.environ.get("VMR_DP_FORCE_COLLECTIVES", "0") != "1"):
        return
    arena = getattr(optimizer, "arena", None)
    inside = set()
    if arena is not None:
        dist.broadcast(arena.flat_p, src)
        inside = {id(p) for p in arena.params}
    rest = [p for p in model.parameters() if id(p) not in inside]
    by_dtype = {}
    for p in rest:
        by_dtype.setdefault(p.dtype, []).append(p)
    for ps in by_dtype.values():
        flat = torch.cat([p.data.reshape(-1) for p in ps])
        dist.broadcast(flat, src)
        o = 0
        for p in ps:
            p.data.copy_(flat[o:o + p.numel()].view_as(p.data))
            o += p.numel()
    if optimizer is not None and hasattr(optimizer, "sync_mirrors"):
        optimizer.sync_mirrors()


class GradReducer:
    """Averages the flat gradient arena across ranks, one stage range at a time.

        red = GradReducer(model, opt)
        ...
        opt.zero_grad()
        red.backward(loss)        # == loss.backward(), cut into stages; each finished stage's range is all-reduced
                                  #    asynchronously while the earlier stages still compute
        red.finish()              # wait for the collectives (gloo: divide by the world size)
        opt.step()

    Before the arena exists (the first step) every existing gradient is reduced through one coalesced buffer.
    Parameters whose grad is None are never touched (they stay None on every rank, as the reference's unused tensors
    do).  `reduce_dtype=torch.bfloat16` halves the bytes on the wire (SURVEY.md 8e allows it): the range is cast
    into a bf16 staging buffer, averaged there and cast back; the default keeps the exact fp32 sum."""

    def __init__(self, model: torch.nn.Module, optimizer, reduce_dtype: torch.dtype = torch.float32):
        self.model, self.opt, self.reduce_dtype = model, optimizer, reduce_dtype
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # rehearsal hook (VMR_DP_FORCE_COLLECTIVES=1 with an initialised 1-rank group): issue every collective although
        # it is an identity -- the only way to put RCCL itself through this call pattern on a one-GPU box
        import os
        self.collect = self.world > 1 or (dist.is_initialized() and os.environ.get("VMR_DP_FORCE_COLLECTIVES", "0") == "1")
        # RCCL averages in the collective itself (ncclAvg): no extra divide pass over the arena; gloo (CPU tests and
        # the one-GPU rehearsal) only sums
        self.avg = dist.is_initialized() and dist.get_backend() == "nccl"
        self.op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.handles = []          # (work handle, fp32 range view, staging buffer or None)
        self.reduced = set()
        self.staging: Optional[torch.Tensor] = None
        self.launch_log = []       # stage indices in launch order (tests)

    # -- the cut backward pass ----------------------------------------------------
    def backward(self, loss: torch.Tensor):
        if hasattr(self.model, "segmented_backward") and getattr(self.model, "backward_cuts", False):
            self.model.segmented_backward(loss, self.stage_done)
        else:
            loss.backward()

    def ranges(self):
        arena = getattr(self.opt, "arena", None)
        return None if arena is None else getattr(arena, "segment_ranges", [(0, arena.flat_g.numel())])

    def stage_done(self, i: int):
        """Every gradient of stage i is final: send its arena range (callable from a captured-graph trainer too:
        it only enqueues a collective, on RCCL's stream, ordered after the current stream)."""
        rg = self.ranges()
        if not self.collect or rg is None or i >= len(rg) or i in self.reduced:
            return
        self.reduced.add(i)
        lo, hi = rg[i]
        if hi > lo:
            self._launch(self.opt.grad_arena[lo:hi])
            self.launch_log.append(i)

    def _launch(self, view: torch.Tensor):
        if self.reduce_dtype == torch.float32:
            self.handles.append((dist.all_reduce(view, op=self.op, async_op=True), view, None))
            return
        if self.staging is None or self.staging.numel() < self.opt.grad_arena.numel():
            self.staging = torch.empty(self.opt.grad_arena.numel(), device=view.device, dtype=self.reduce_dtype)
        off = view.storage_offset() - self.opt.grad_arena.storage_offset()
        st = self.staging[off:off + view.numel()]
        st.copy_(view)
        self.handles.append((dist.all_reduce(st, op=self.op, async_op=True), view, st))

    def finish(self):
        """Call after backward(): reduce whatever stage has not been sent yet, wait, average."""
        if not self.collect:
            return
        rg = self.ranges()
        if rg is None:                 # first step: the arena is not built yet -- one coalesced buffer
            grads = [p.grad for p in self.model.parameters() if p.grad is not None]
            if grads:
                flat = torch.cat([g.reshape(-1).float() for g in grads])
                dist.all_reduce(flat, op=self.op)
                if not self.avg:
                    flat.div_(self.world)
                o = 0
                for g in grads:
                    g.copy_(flat[o:o + g.numel()].view_as(g))
                    o += g.numel()
            return
        for i in range(len(rg) - 1, -1, -1):
            self.stage_done(i)
        for h, view, st in self.handles:
            h.wait()
            if st is not None:
                view.copy_(st)
            if not self.avg:
                view.div_(self.world)
        self.handles.clear()
        self.reduced.clear()


class ShardedReducer(GradReducer):
    """Reduce-scatter + sharded AdamW + all-gather instead of all-reduce + replicated AdamW.

    optim.FlatArena lays every stage out as [matrix region | fp32 region].  Per finished stage of the backward pass:
      * the matrix region's gradients (the GEMM weights: 99 % of the arena) are REDUCE-SCATTERED -- rank r ends up with the
        average of its 1/N slice only, (N-1)/N x bytes on the wire instead of all-reduce's 2(N-1)/N;
      * the small fp32 region (biases, LayerNorm / depthwise parameters, tables read in fp32) is all-reduced: every rank
        updates it, so the fp32 masters the kernels read stay current everywhere.
    FlatAdamW then updates only this rank's slices (1/N of the optimizer's HBM traffic: `opt.step()` runs in three
    parts with two host hops -- the clip norm needs one scalar all-reduce over the ranks' partial sums of squares) and the
    16-bit mirrors of the matrix regions are ALL-GATHERED (half the bytes of the gradients) before the K-major copies are
    refreshed.  fp32 masters of foreign slices go stale by design: `gather_masters()` before a checkpoint.

    Results: the mirrors every rank computes with, and the owner's masters, are bit-identical to the all-reduce path
    whenever the backend's reduce-scatter and all-reduce add in the same order (always at world size 2).
    World sizes must divide 64 / 8 (2, 4, 8): the regions are padded to 64 elements."""

    def __init__(self, model, optimizer, reduce_dtype: torch.dtype = torch.float32):
        super().__init__(model, optimizer, reduce_dtype)
        assert reduce_dtype == torch.float32, "the sharded path reduces in fp32 (the gather already moves 16-bit mirrors)"
        assert self.world in (1, 2, 4, 8), "ShardedReducer: world size must be 1, 2, 4 or 8"
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self._native = None         # does the backend take reduce_scatter_tensor / all_gather_into_tensor on this device?
        optimizer.shard = self

    # -- layout -------------------------------------------------------------------
    def split(self):
        arena = getattr(self.opt, "arena", None)
        return None if arena is None else arena.segment_split

    def my_slices(self):
        """[(lo, hi)] of this rank's slice of every stage's matrix region."""
        out = []
        for lo, mid, _ in self.split():
            sz = (mid - lo) // self.world
            out.append((lo + self.rank * sz, lo + (self.rank + 1) * sz))
        return out

    def fp32_regions(self):
        return [(mid, hi) for _, mid, hi in self.split()]

    # -- collectives ----------------------------------------------------------------
    def _probe(self, like: torch.Tensor):
        if self._native is None:
            try:        # (gloo moves CPU tensors through these; with CUDA tensors it may not)
                a = torch.zeros(8 * self.world, device=like.device)
                b = torch.zeros(8, device=like.device)
                dist.reduce_scatter_tensor(b, a, op=dist.ReduceOp.SUM)
                dist.all_gather_into_tensor(a, b)
                self._native = True
            except Exception:
                self._native = False
        return self._native

    def stage_done(self, i: int):
        sp = self.split()
        if not self.collect or sp is None or i >= len(sp) or i in self.reduced:
            return
        self.reduced.add(i)
        lo, mid, hi = sp[i]
        g = self.opt.grad_arena
        if mid > lo:
            sz = (mid - lo) // self.world
            mine = g[lo + self.rank * sz: lo + (self.rank + 1) * sz]
            if self._probe(g):
                if self.avg:        # RCCL: in place (the output is the rank's own slice of the input), averaged by the collective
                    self.handles.append((dist.reduce_scatter_tensor(mine, g[lo:mid], op=self.op, async_op=True), mine, None))
                else:
                    out = torch.empty_like(mine)
                    self.handles.append((dist.reduce_scatter_tensor(out, g[lo:mid], op=self.op, async_op=True), mine, out))
            else:                   # no reduce-scatter for this device in the backend: the same sums through all-reduce
                self.handles.append((dist.all_reduce(g[lo:mid], op=self.op, async_op=True), g[lo:mid], None))
        if hi > mid:
            self.handles.append((dist.all_reduce(g[mid:hi], op=self.op, async_op=True), g[mid:hi], None))
        self.launch_log.append(i)

    def finish(self):
        if not self.collect:
            return
        if self.split() is None:
            return super().finish()         # first step, no arena yet: one coalesced all-reduce
        for i in range(len(self.split()) - 1, -1, -1):
            self.stage_done(i)
        for h, view, st in self.handles:
            h.wait()
            if st is not None:
                view.copy_(st)
            if not self.avg:
                view.div_(self.world)
        self.handles.clear()
        self.reduced.clear()

    def sum_scalar(self, t: torch.Tensor):
        """partial sums of squares -> their sum over the ranks (the clip norm of the averaged gradient)"""
        if self.collect:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def gather(self, flat: torch.Tensor):
        """All-gather every stage's matrix region of `flat` (the 16-bit mirror arena, or the fp32 masters): each rank
        contributes its own slice."""
        if not self.collect:
            return
        hs = []
        for (lo, mid, _), (a, b) in zip(self.split(), self.my_slices()):
            if mid == lo:
                continue
            if self._probe(flat):
                src = flat[a:b] if self.avg else flat[a:b].clone()    # (RCCL gathers in place)
                hs.append(dist.all_gather_into_tensor(flat[lo:mid], src, async_op=True))
            else:
                sz = b - a
                for r in range(self.world):
                    hs.append(dist.broadcast(flat[lo + r * sz: lo + (r + 1) * sz], r, async_op=True))
        for h in hs:
            h.wait()

    def gather_masters(self):
        """Bring the fp32 masters of the foreign slices up to date (before state_dict() / a checkpoint)."""
        self.gather(self.opt.arena.flat_p)
