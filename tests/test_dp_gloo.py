"""CPU, world_size 2, gloo: the N>1 path (clip sharding, parameter broadcast, gradient averaging over the flat arena
one stage range at a time with the backward pass cut at the stage boundaries) and bench.py's own rank launcher."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


class Toy(torch.nn.Module):
    """Two stages (a | b + layer_norm) with the model-side protocol of SeqPAN: param_segment, backward_cuts,
    backward_plan / segmented_backward."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 4)
        self.unused = torch.nn.Parameter(torch.ones(3))          # never gets a gradient
        self.layer_norm = torch.nn.LayerNorm(4)
        self.backward_cuts = False
        self._cut = None

    def param_segment(self, name):
        return 0 if name.startswith("a.") else 1

    def forward(self, x):
        h = torch.relu(self.a(x))
        if self.backward_cuts:
            leaf = h.detach().requires_grad_(True)
            self._cut = (h, leaf)
            h = leaf
        return self.layer_norm(self.b(h)).sum(-1)

    def segmented_backward(self, loss, after_stage=None):
        loss.backward()
        if after_stage:
            after_stage(1)
        h, leaf = self._cut
        h.backward(leaf.grad)
        if after_stage:
            after_stage(0)


class ToyOpt:
    """Stand-in for FlatAdamW on CPU: same arena protocol, SGD update."""

    def __init__(self, model):
        self.model, self.arena = model, None

    grad_arena = property(lambda s: None if s.arena is None else s.arena.flat_g)
    names = property(lambda s: s.arena.names)
    offsets = property(lambda s: s.arena.offsets)

    def zero_grad(self):
        if self.arena is None:
            for p in self.model.parameters():
                p.grad = None
        else:
            self.arena.flat_g.zero_()

    def step(self):
        from vmrframe_amd.optim import FlatArena
        if self.arena is None:
            self.arena = FlatArena(self.model)
        self.arena.flat_p.add_(self.arena.flat_g, alpha=-0.1)


def _worker(rank, world, port, q, cuts=True, reduce_dtype=torch.float32):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from vmrframe_amd import dp
    dp.init_process_group_from_env("gloo")
    torch.manual_seed(100 + rank)                 # different init per rank -> broadcast must fix it
    model = Toy()
    dp.broadcast_parameters(model)
    torch.manual_seed(7)
    full = {"x": torch.randn(12, 8), "y": torch.randn(12)}
    mine = dp.shard_batch(full, rank, world)
    assert mine["x"].shape[0] == 6
    opt = ToyOpt(model)
    model.backward_cuts = cuts
    red = dp.GradReducer(model, opt, reduce_dtype=reduce_dtype)
    ref = Toy(); ref.load_state_dict(model.state_dict())
    tol = 1e-6 if reduce_dtype == torch.float32 else 2e-2
    for it in range(4):
        loss = ((model(mine["x"]) - mine["y"]) ** 2).mean()
        opt.zero_grad(); red.backward(loss); red.finish()
        # reference: mean over ranks of per-rank gradients == gradient of the mean of the two shard losses
        rl = sum(((ref(full["x"][r::world]) - full["y"][r::world]) ** 2).mean() for r in range(world)) / world
        for p in ref.parameters():
            p.grad = None
        rl.backward()
        for (n, p), (_, rp) in zip(model.named_parameters(), ref.named_parameters()):
            if rp.grad is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
            else:
                assert torch.allclose(p.grad, rp.grad, atol=tol, rtol=tol), (it, n)
        opt.step()
        with torch.no_grad():
            for rp in ref.parameters():
                if rp.grad is not None:
                    rp.add_(rp.grad, alpha=-0.1)
    assert model.unused.grad is None
    # once the arena exists (steps 1..3) the stage ranges go out last stage first when the pass is cut
    assert red.launch_log == ([1, 0] * 3 if cuts else [1, 0] * 3), red.launch_log
    assert opt.arena.segment_ranges[0][1] == opt.arena.segment_ranges[1][0] and opt.arena.names[0].startswith("a.")
    nb = len(opt.arena.segment_ranges)
    sd = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    gathered = [torch.zeros_like(sd) for _ in range(world)]
    dist.all_gather(gathered, sd)
    assert torch.allclose(gathered[0], gathered[1], atol=0 if reduce_dtype == torch.float32 else 1e-2)   # lock-step
    q.put((rank, nb))
    dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("cuts,reduce_dtype", [(True, torch.float32), (False, torch.float32), (True, torch.bfloat16)])
def test_two_rank_gradient_averaging_gloo(cuts, reduce_dtype):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, cuts, reduce_dtype)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    res = sorted(q.get() for _ in range(2))
    assert [r for r, _ in res] == [0, 1]
    assert res[0][1] == 2                     # the arena is laid out in the model's two stages


@pytest.mark.timeout(240)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it: the parent spawns one process per rank (before touching
    any GPU), the ranks rendezvous on 127.0.0.1 and rank 0 prints the one JSON line.  --selftest-launcher stops each
    rank after its first collective (gloo), so this runs on a GPU-less host."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launcher"],
                         env=env, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # ONE line, from rank 0 only
    d = json.loads(lines[0])
    assert d["world_observed"] == 2 and d["sum_of_ranks"] == 3.0
    # under an external launcher (torch.distributed.run exports WORLD_SIZE) it must NOT spawn again
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--selftest-launcher"],
                         env=env2, capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])["world_observed"] == 1
