"""Batch data parallelism for the SeqPAN path: one process per GPU, each rank
trains on its own shard of clips and gradients are averaged with RCCL
all-reduce over xGMI (torch.distributed backend "nccl" == RCCL on ROCm).

Replaces the reference's single-process nn.DataParallel (reference main.py:22-24),
which cannot work for this model.  Semantics are the
reference's per-replica forward at the local batch (the batch-axis attention of
layers.py:567-574 sees the LOCAL batch) with DDP-style gradient averaging.

Gradients live in one flat fp32 arena (vmrframe_amd/optim.py); it is cut into a
few large buckets that are all-reduced asynchronously as soon as autograd has
produced every gradient of the bucket (post-accumulate hooks), so the transfer
overlaps the rest of the backward.  Point-to-point xGMI favours few, large
messages: ~32 MiB buckets.
"""
from __future__ import annotations

from typing import Dict, List, Optional

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
    if not dist.is_initialized() or dist.get_world_size() == 1:
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
    """Averages the flat gradient arena across ranks, bucket by bucket, overlapped with
    backward.  Before the arena exists (first step) it falls back to per-tensor all-reduce
    of whatever gradients exist.  Parameters whose grad is None are skipped (they stay None
    on every rank, as the reference's unused tensors do)."""

    def __init__(self, model: torch.nn.Module, optimizer, bucket_bytes: int = 32 << 20, use_hooks: bool = True):
        # use_hooks=False: no autograd hooks at all -- finish() all-reduces the arena in bucket-sized
        # chunks after backward.  Required when backward is replayed from a captured HIP graph (hooks do
        # not run on replay, and a collective must never be issued while a graph is being captured).
        self.model, self.opt, self.bucket_bytes, self.use_hooks = model, optimizer, bucket_bytes, use_hooks
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # RCCL averages in the collective itself (ncclAvg): no extra 262 MB divide pass over the arena; gloo (CPU
        # tests) only sums
        self.avg = dist.is_initialized() and dist.get_backend() == "nccl"
        self.op = dist.ReduceOp.AVG if self.avg else dist.ReduceOp.SUM
        self.buckets: List[dict] = []
        self.param_bucket = {}
        self.handles = []
        self.hooks = []

    def _setup_buckets(self):
        opt = self.opt
        flat = opt.grad_arena
        per = max(1, self.bucket_bytes // 4)
        # cut in REVERSE registration order ~ the order gradients become ready
        names = list(opt.names)[::-1]
        params = dict(self.model.named_parameters())
        cur = None
        for n in names:
            o, k = opt.offsets[n], params[n].numel()
            if cur is None or (cur["hi"] - o) > per:
                cur = {"lo": o, "hi": o + (k + 7) // 8 * 8, "pending": 0, "count": 0}
                self.buckets.append(cur)
            cur["lo"] = o
            cur["count"] += 1
            self.param_bucket[n] = cur
        for n in names:
            b = self.param_bucket[n]
            self.hooks.append(params[n].register_post_accumulate_grad_hook(self._make_hook(b)))
        for b in self.buckets:
            b["view"] = flat[b["lo"]:b["hi"]]
            b["pending"] = b["count"]

    def _make_hook(self, bucket):
        def hook(_p):
            bucket["pending"] -= 1
            if bucket["pending"] == 0:
                self._launch(bucket)
        return hook

    def _launch(self, bucket):
        if self.world > 1:
            self.handles.append(dist.all_reduce(bucket["view"], op=self.op, async_op=True))
        bucket["launched"] = True

    def finish(self):
        """Call after backward(): wait for the buckets, average, re-arm."""
        if self.world == 1:
            return
        if self.opt.grad_arena is None:            # first step: arena not built yet
            for p in self.model.parameters():
                if p.grad is not None:
                    dist.all_reduce(p.grad, op=self.op)
                    if not self.avg:
                        p.grad.div_(self.world)
            return
        if not self.use_hooks:
            flat = self.opt.grad_arena
            per = max(1, self.bucket_bytes // 4)
            hs = [dist.all_reduce(flat[o:o + per], op=self.op, async_op=True)
                  for o in range(0, flat.numel(), per)]
            for h in hs:
                h.wait()
            if not self.avg:
                flat.div_(self.world)
            return
        if not self.buckets:
            # arena was just built by the previous optimizer step but hooks were not armed for
            # this backward: reduce the whole arena in one go, then arm the hooks.
            dist.all_reduce(self.opt.grad_arena, op=self.op)
            if not self.avg:
                self.opt.grad_arena.div_(self.world)
            self._setup_buckets()
            return
        for b in self.buckets:                     # parameters that got no gradient this step
            if not b.get("launched"):
                self._launch(b)
        for h in self.handles:
            h.wait()
        self.handles.clear()
        if not self.avg:
            self.opt.grad_arena.div_(self.world)
        for b in self.buckets:
            b["pending"] = b["count"]
            b["launched"] = False
