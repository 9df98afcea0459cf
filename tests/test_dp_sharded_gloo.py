"""CPU, world_size 2, gloo: the reduce-scatter + sharded optimizer + all-gather protocol (dp.ShardedReducer)
against the all-reduce + replicated optimizer path (dp.GradReducer) on the SAME per-rank gradients.

What is pinned: the arena's [matrix region | fp32 region] layout per stage with 64-element region bounds; every rank's
16-bit mirror, its own slices of the fp32 masters and the whole fp32 regions come out bit-identical to the all-reduce
path (at world size 2 a sum of two addends has one order); foreign master slices are stale until gather_masters();
training through the mirrors stays in lock-step across the ranks."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _mirrored(p):
    """value from the 16-bit mirror (what the GEMM kernels read), gradient to the fp32 master"""
    w = getattr(p, "_vmr_w16", None)
    return p if w is None else w.float() + (p - p.detach())


class Toy(torch.nn.Module):
    """Two stages; matrices are consumed through their mirrors, vectors in fp32 -- the product path's split."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(64, 128)
        self.b = torch.nn.Linear(128, 64)
        self.layer_norm = torch.nn.LayerNorm(64)
        self.table = torch.nn.Parameter(torch.randn(64, 64) * 0.1)      # a matrix some kernel reads in fp32
        self.backward_cuts = False
        self._cut = None

    def param_segment(self, name):
        return 0 if name.startswith("a.") else 1

    def fp32_consumed(self):
        return ["table"]

    def forward(self, x):
        h = torch.relu(torch.nn.functional.linear(x, _mirrored(self.a.weight), self.a.bias))
        if self.backward_cuts:
            leaf = h.detach().requires_grad_(True)
            self._cut = (h, leaf)
            h = leaf
        y = torch.nn.functional.linear(h, _mirrored(self.b.weight), self.b.bias)
        return (self.layer_norm(y) @ self.table).sum(-1)

    def segmented_backward(self, loss, after_stage=None):
        loss.backward()
        if after_stage:
            after_stage(1)
        h, leaf = self._cut
        h.backward(leaf.grad)
        if after_stage:
            after_stage(0)


class ToyOpt:
    """FlatAdamW's arena protocol on the CPU with an SGD update: replicated, or on this rank's slices (shard set)."""

    def __init__(self, model):
        self.model, self.arena, self.shard = model, None, None
        self.gnorm_sq = torch.zeros(1)

    grad_arena = property(lambda s: None if s.arena is None else s.arena.flat_g)

    def build(self):
        from vmrframe_amd.optim import FlatArena
        if self.arena is None:
            self.arena = FlatArena(self.model, mirror_dtype=torch.bfloat16)

    def _sgd(self, lo, hi):
        A = self.arena
        A.flat_p[lo:hi].add_(A.flat_g[lo:hi], alpha=-0.1)
        A.flat_w[lo:hi].copy_(A.flat_p[lo:hi])

    def step(self):
        self.build()
        A = self.arena
        if self.shard is None:
            self._sgd(0, A.flat_p.numel())
            return
        for lo, hi in self.shard.my_slices() + self.shard.fp32_regions():
            self._sgd(lo, hi)
        self.shard.gather(A.flat_w)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from vmrframe_amd import dp
    dp.init_process_group_from_env("gloo")
    torch.manual_seed(3)
    model = Toy()
    torch.manual_seed(7)
    full = {"x": torch.randn(12, 64), "y": torch.randn(12)}
    mine = dp.shard_batch(full, rank, world)
    opt = ToyOpt(model)
    ((model(mine["x"]) - mine["y"]) ** 2).mean().backward()      # the arena learns which parameters get gradients
    opt.build()
    A = opt.arena
    # ---- layout: [matrix region | fp32 region] per stage, 64-element bounds, the declared fp32 consumer on the fp32 side
    assert len(A.segment_split) == 2
    for lo, mid, hi in A.segment_split:
        assert lo % 64 == 0 and mid % 64 == 0 and hi % 64 == 0 and lo < mid < hi
    off = A.offsets
    (lo0, mid0, hi0), (lo1, mid1, hi1) = A.segment_split
    assert lo0 <= off["a.weight"] < mid0 <= off["a.bias"] < hi0
    assert lo1 <= off["b.weight"] < mid1 and mid1 <= off["table"] < hi1 and mid1 <= off["layer_norm.weight"] < hi1
    model.backward_cuts = True
    ar = dp.GradReducer(model, opt)
    sh = dp.ShardedReducer(model, opt)
    opt.shard = None
    assert sh.my_slices() == [(lo0 + rank * (mid0 - lo0) // 2, lo0 + (rank + 1) * (mid0 - lo0) // 2),
                              (lo1 + rank * (mid1 - lo1) // 2, lo1 + (rank + 1) * (mid1 - lo1) // 2)]
    for it in range(3):
        # ONE backward pass; its local gradients go through both paths from the same state
        A.flat_g.zero_()
        loss = ((model(mine["x"]) - mine["y"]) ** 2).mean()
        model.segmented_backward(loss)
        g_local = A.flat_g.clone()
        p0, w0 = A.flat_p.clone(), A.flat_w.clone()
        # path A: all-reduce + replicated update
        opt.shard = None
        for i in (1, 0):
            ar.stage_done(i)
        ar.finish()
        g_avg = A.flat_g.clone()
        opt.step()
        pA, wA = A.flat_p.clone(), A.flat_w.clone()
        # path B: reduce-scatter / all-reduce per region + update of the local slices + gather of the mirrors
        A.flat_p.copy_(p0); A.flat_w.copy_(w0); A.flat_g.copy_(g_local)
        opt.shard = sh
        for i in (1, 0):
            sh.stage_done(i)
        sh.finish()
        for lo, hi in sh.my_slices() + sh.fp32_regions():
            assert torch.equal(A.flat_g[lo:hi], g_avg[lo:hi]), (it, lo, hi)     # the averaged gradient where this rank needs it
        opt.step()
        assert torch.equal(A.flat_w, wA), it                                     # every mirror, on every rank
        for lo, hi in sh.my_slices() + sh.fp32_regions():
            assert torch.equal(A.flat_p[lo:hi], pA[lo:hi]), (it, lo, hi)         # own master slices + the fp32 regions
        other = [(lo + (1 - rank) * (mid - lo) // 2, lo + (2 - rank) * (mid - lo) // 2) for lo, mid, _ in A.segment_split]
        assert any(not torch.equal(A.flat_p[lo:hi], pA[lo:hi]) for lo, hi in other)   # foreign slices: stale by design ...
        sh.gather_masters()
        assert torch.equal(A.flat_p, pA), it                                     # ... until gathered (checkpoint time)
    assert sh.launch_log == [1, 0] * 3
    sd = A.flat_w.float()
    gathered = [torch.zeros_like(sd) for _ in range(world)]
    dist.all_gather(gathered, sd)
    assert torch.equal(gathered[0], gathered[1])                                 # lock-step
    q.put(rank)
    dist.destroy_process_group()


@pytest.mark.timeout(240)
def test_sharded_optimizer_equals_all_reduce_path_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(150)
        assert p.exitcode == 0
    assert sorted(q.get() for _ in range(2)) == [0, 1]
