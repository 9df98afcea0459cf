"""GPU: the data-parallel path of the REAL model -- the backward pass cut into stages, the
piecewise-graph trainer that issues one all-reduce per finished stage, and a 2-rank run (gloo, both ranks on this one
GPU: RCCL refuses two ranks on one device, and the boxes here have one) checked against the mean of the per-shard
gradients computed by a single process.  The per-rank forward is the reference at the LOCAL batch (the batch-axis
attention of layers.py:567-574 sees the local batch), so that -- not a full-batch run -- is the expected value."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from tests.helpers import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda")


def _build(name, dtype, dev, seed=77):
    from tests.test_gpu_trainer import build
    z, cfg, batch, g, weights = load_golden(name)
    return build(cfg, weights, dtype, dev, g, seed=seed), cfg, batch, g, weights


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


class FakeReducer:
    """world = 2 without a process group: records what the trainer hands to RCCL, reduces nothing."""

    def __init__(self, model, opt):
        self.model, self.opt, self.world, self.log = model, opt, 2, []

    def ranges(self):
        return self.opt.arena.segment_ranges

    def backward(self, loss):
        self.model.segmented_backward(loss, self.stage_done)

    def stage_done(self, i):
        self.log.append(i)

    def finish(self):
        self.log.append("finish")


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_stage_cut_backward_equals_plain_backward(dev, dtype):
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    m, cfg, batch, g, weights = _build("g_small", dtype, dev)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    opt = FlatAdamW(m, lr=0.0, max_norm=1.0)
    grads = {}
    for cuts in (False, True, False, True):          # passes 0-1 build the arena; 2-3 are compared
        m.backward_cuts = cuts
        loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
        opt.zero_grad()
        order = []
        if cuts:
            m.segmented_backward(loss, order.append)
            assert order == [4, 3, 2, 1, 0]
        else:
            loss.backward()
        opt.step()
        grads[cuts] = opt.arena.flat_g.clone()
    tol = 1e-5 if dtype == "fp32" else 5e-2            # (bf16: atomics-order noise of the forward, see test_gpu_trainer)
    assert _rel(grads[True], grads[False]) < tol
    # the arena is laid out stage by stage, ranges contiguous and ordered like the forward
    rg = opt.arena.segment_ranges
    assert len(rg) == 5 and rg[0][0] == 0 and rg[-1][1] == opt.arena.flat_g.numel()
    assert all(a[1] == b[0] for a, b in zip(rg[:-1], rg[1:]))
    for n in opt.names:
        lo, hi = rg[m.param_segment(n)]
        assert lo <= opt.offsets[n] < hi, n
    frac = [(hi - lo) / rg[-1][1] for lo, hi in rg]
    print("[stages] share of the arena per stage:", [round(f, 3) for f in frac])


def test_piecewise_graph_trainer_equals_single_graph(dev):
    """N > 1 trainer (one graph per backward stage + an optimizer graph, collectives between replays) against the
    single-graph trainer, 3 steps each from the same state, bf16, dropout off: same weights afterwards, and the
    stages are handed over last-first, each right after its own graph."""
    import vmrframe_amd as V
    from vmrframe_amd.optim import FlatAdamW
    from vmrframe_amd.trainer import GraphedTrainStep
    sched = dict(warmup_steps=0.0, total_steps=10)
    A, cfg, batch, g, weights = _build("g_small", "bf16", dev)
    optA = FlatAdamW(A, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
    red = FakeReducer(A, optA)
    stepA = GraphedTrainStep(A, optA, V.train_engine_SeqPAN, cfg, red, warmup=2).capture(batch)
    assert len(stepA.pieces) == 5 and stepA.g_opt is not None and A.backward_cuts
    B, _, _, _, _ = _build("g_small", "bf16", dev)
    optB = FlatAdamW(B, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
    stepB = GraphedTrainStep(B, optB, V.train_engine_SeqPAN, cfg, None, warmup=2).capture(batch)
    assert len(stepB.pieces) == 1
    for it in range(3):
        for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v),
                         (optA.step_t, optB.step_t)):
            dst.copy_(src)
        optA.sync_mirrors()
        before = optB.arena.flat_p.clone()
        red.log.clear()
        la, lb = float(stepA().item()), float(stepB().item())
        torch.cuda.synchronize()
        assert red.log == [4, 3, 2, 1, 0, "finish"]
        assert abs(la - lb) < 2e-3 * max(1.0, abs(lb)), (it, la, lb)
        dA, dB = (optA.arena.flat_p - before).double(), (optB.arena.flat_p - before).double()
        keep = optB.arena.flat_g.abs() >= 1e-4 * float(optB.arena.flat_g.abs().max())   # (noise-only gradients: see test_gpu_trainer)
        r = float((dA - dB)[keep].norm() / dB[keep].norm())
        assert r < 3e-2, (it, r)
    assert int(optA.step_t.item()) == int(optB.step_t.item()) == 5


# --------------------------------------------------------------------------------------------------------------------
# two ranks (gloo) on this GPU, real model + FlatAdamW + stage-cut overlap protocol
# --------------------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank(rank, world, port, outdir, graph):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import vmrframe_amd as V
    from vmrframe_amd import dp
    from vmrframe_amd.optim import FlatAdamW
    from vmrframe_amd.trainer import GraphedTrainStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dp.init_process_group_from_env("gloo")
    torch.manual_seed(100 + rank)                           # different init per rank: the broadcast must fix it
    m, cfg, batch, g, weights = _build("g_small", "bf16", dev)
    if rank == 1:
        with torch.no_grad():
            for p in m.parameters():
                if p.requires_grad:
                    p.add_(0.01)
    dp.broadcast_parameters(m)
    shard = dp.shard_batch(batch, rank, world)
    m.gumbel_override = g[rank::world].to(dev)
    m.backward_cuts = True
    opt = FlatAdamW(m, lr=0.0, max_norm=1.0)
    red = dp.GradReducer(m, opt)
    if graph:
        step = GraphedTrainStep(m, opt, V.train_engine_SeqPAN, cfg, red, warmup=2).capture(shard)
        step()
    else:
        dshard = {k: v.to(dev) for k, v in shard.items()}
        for _ in range(3):
            loss, _ = V.train_engine_SeqPAN(m, dshard, cfg, "train")
            opt.zero_grad(); red.backward(loss); red.finish(); opt.step()
    torch.cuda.synchronize()
    assert red.launch_log[-5:] == [4, 3, 2, 1, 0]
    torch.save({"g": opt.arena.flat_g.cpu(), "names": opt.names, "offsets": opt.offsets,
                "w": m.P("dual_attention_block_1.dense_1.conv1d.weight").detach().cpu()},
               os.path.join(outdir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("graph", [False, True])
def test_two_ranks_real_model_gradient_average(dev, tmp_path, graph):
    import vmrframe_amd as V
    from vmrframe_amd import dp
    from vmrframe_amd.optim import FlatAdamW
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, str(tmp_path), graph)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
        assert p.exitcode == 0
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]
    assert torch.equal(got[0]["g"], got[1]["g"])                       # both ranks hold the same averaged gradients
    assert torch.equal(got[0]["w"], got[1]["w"])                       # ... and rank 0's weights
    # expected: mean over ranks of the single-process gradient on each rank's shard
    m, cfg, batch, g, weights = _build("g_small", "bf16", dev)
    opt = FlatAdamW(m, lr=0.0, max_norm=1.0)
    per = []
    for r in range(2):
        shard = {k: v.to(dev) for k, v in dp.shard_batch(batch, r, 2).items()}
        m.gumbel_override = g[r::2].to(dev)
        for _ in range(2):
            loss, _ = V.train_engine_SeqPAN(m, shard, cfg, "train")
            opt.zero_grad(); loss.backward(); opt.step()
        per.append(opt.arena.flat_g.cpu().clone())
    assert opt.names == got[0]["names"]
    want = (per[0] + per[1]) / 2
    r_ = _rel(got[0]["g"], want)
    print(f"[2 ranks, graph={graph}] averaged arena vs mean of shard gradients: rel {r_:.3e}")
    assert r_ < 5e-2                                                    # bf16 forward noise; a missing average is 0.5-1
    assert torch.equal(got[0]["w"], m.P("dual_attention_block_1.dense_1.conv1d.weight").detach().cpu())


# --------------------------------------------------------------------------------------------------------------------
# reduce-scatter + sharded FlatAdamW + all-gather of the mirrors (dp.ShardedReducer) against all-reduce + replicated AdamW
# --------------------------------------------------------------------------------------------------------------------
def _rank_sharded(rank, world, port, outdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import vmrframe_amd as V
    from vmrframe_amd import dp
    from vmrframe_amd.optim import FlatAdamW
    from vmrframe_amd.trainer import GraphedTrainStep
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dp.init_process_group_from_env("gloo")
    m, cfg, batch, g, weights = _build("g_cfg1", "bf16", dev)      # D = 512: the projections are GEMM matrices (>= 64 wide)
    dp.broadcast_parameters(m)
    shard = {k: v.to(dev) for k, v in dp.shard_batch(batch, rank, world).items()}
    m.gumbel_override = g[rank::world].to(dev)
    m.backward_cuts = True
    # max_norm far above the gradient norm: the clip factor is exactly 1 on both paths (the two paths add the squares in
    # different orders, so an ACTIVE clip would differ in the last bit)
    opt = FlatAdamW(m, lr=1e-3, weight_decay=0.01, max_norm=1e9)
    ar = dp.GradReducer(m, opt)
    loss, _ = V.train_engine_SeqPAN(m, shard, cfg, "train")
    opt.zero_grad(); ar.backward(loss); ar.finish(); opt.step()            # builds the arenas
    A = opt.arena
    sh = dp.ShardedReducer(m, opt)
    opt.shard = None
    nst = len(A.segment_split)
    assert all(lo % 64 == 0 and mid % 64 == 0 and hi % 64 == 0 for lo, mid, hi in A.segment_split)
    mat = sum(mid - lo for lo, mid, _ in A.segment_split)
    assert mat > 0.9 * A.flat_p.numel()                                    # the GEMM weights are the arena
    for n in m.fp32_consumed():
        st = m.param_segment(n)
        assert A.segment_split[st][1] <= A.offsets[n] < A.segment_split[st][2], n
    res = {}
    for it in range(2):
        loss, _ = V.train_engine_SeqPAN(m, shard, cfg, "train")
        opt.zero_grad()
        m.segmented_backward(loss)
        m._cache.state.flush_reduce(); m._cache.state.flush_colreduce()
        torch.cuda.synchronize()
        snap = [t.clone() for t in (A.flat_g, A.flat_p, A.flat_w, A.flat_wt, opt.m, opt.v, opt.step_t)]
        host_t = opt.t
        # path A
        opt.shard = None
        for i in range(nst - 1, -1, -1):
            ar.stage_done(i)
        ar.finish(); opt.step()
        torch.cuda.synchronize()
        wA, wtA, pA, mA, vA = A.flat_w.clone(), A.flat_wt.clone(), A.flat_p.clone(), opt.m.clone(), opt.v.clone()
        # path B from the same state and the same local gradients
        for dst, src in zip((A.flat_g, A.flat_p, A.flat_w, A.flat_wt, opt.m, opt.v, opt.step_t), snap):
            dst.copy_(src)
        opt.t = host_t
        opt.shard = sh
        for i in range(nst - 1, -1, -1):
            sh.stage_done(i)
        sh.finish(); opt.step()
        torch.cuda.synchronize()
        assert torch.equal(A.flat_w, wA) and torch.equal(A.flat_wt, wtA), it      # what every rank computes with
        for lo, hi in sh.my_slices() + sh.fp32_regions():
            assert torch.equal(A.flat_p[lo:hi], pA[lo:hi]), (it, lo, hi)
        for lo, hi in sh.my_slices() + sh.fp32_regions():          # Adam moments: kept for the own slices only
            assert torch.equal(opt.m[lo:hi], mA[lo:hi]) and torch.equal(opt.v[lo:hi], vA[lo:hi]), (it, lo, hi)
        sh.gather_masters()
        assert torch.equal(A.flat_p, pA), it
        opt.m.copy_(mA); opt.v.copy_(vA)     # (the next round's path A needs the moments of every slice)
        res[it] = A.flat_w.cpu()
    # the captured multi-rank step with the sharded optimizer: three optimizer graphs around two exchanges
    step = GraphedTrainStep(m, opt, V.train_engine_SeqPAN, cfg, sh, warmup=2).capture(shard)
    assert step.sharded and len(step.g_parts) == 3 and step.g_opt is None
    # a full, rank-independent snapshot of the optimizer state (foreign slices of the masters and of the Adam moments are
    # stale by design: gather them)
    sh.gather_masters(); sh.gather(opt.m); sh.gather(opt.v)
    torch.cuda.synchronize()
    w_start = A.flat_w.clone()
    p_full, m_start, v_start, t_start = A.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.step_t.clone()
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    w_sharded = A.flat_w.clone()
    # the same two steps on the all-reduce path from the same state (a second model: the first one's graphs are captured):
    # a kernel that read a STALE fp32 master on the rank that does not own it (e.g. a weight cast from the master instead of
    # taken from the mirror) would show up here as a different forward pass
    m2, _, _, _, _ = _build("g_cfg1", "bf16", dev)
    m2.gumbel_override = g[rank::world].to(dev)
    m2.backward_cuts = True
    opt2 = FlatAdamW(m2, lr=1e-3, weight_decay=0.01, max_norm=1e9)
    ar2 = dp.GradReducer(m2, opt2)
    loss, _ = V.train_engine_SeqPAN(m2, shard, cfg, "train")
    opt2.zero_grad(); ar2.backward(loss); ar2.finish(); opt2.step()
    assert opt2.names == opt.names
    opt2.arena.flat_p.copy_(p_full); opt2.m.copy_(m_start); opt2.v.copy_(v_start); opt2.step_t.copy_(t_start)
    opt2.t = int(t_start.item())
    opt2.sync_mirrors()
    assert torch.equal(opt2.arena.flat_w, w_start)
    step2 = GraphedTrainStep(m2, opt2, V.train_engine_SeqPAN, cfg, ar2, warmup=2)
    # (capture() runs `warmup` eager steps first: restore the state after them)
    step2.capture(shard)
    opt2.arena.flat_p.copy_(p_full); opt2.m.copy_(m_start); opt2.v.copy_(v_start); opt2.step_t.copy_(t_start)
    opt2.sync_mirrors()
    for _ in range(2):
        step2()
    torch.cuda.synchronize()
    d = (opt2.arena.flat_w.float() - w_sharded.float()).abs()
    moved = (w_sharded.float() - w_start.float()).abs()
    frac_equal = float((d == 0).float().mean())
    assert float(moved.max()) > 0 and frac_equal > 0.995 and float(d.max()) <= 2 * float(moved.max()) * 0.02 + 1e-3, (frac_equal, float(d.max()))
    torch.save({"w": res, "wg": A.flat_w.cpu(), "logits": step.out["slogits"].float().cpu(), "frac_equal": frac_equal},
               os.path.join(outdir, f"sh{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_sharded_optimizer_equals_all_reduce(dev, tmp_path):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_rank_sharded, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(500)
        assert p.exitcode == 0
    got = [torch.load(os.path.join(tmp_path, f"sh{r}.pt")) for r in range(2)]
    for it in (0, 1):
        assert torch.equal(got[0]["w"][it], got[1]["w"][it])           # the same mirrors on both ranks
    assert torch.equal(got[0]["wg"], got[1]["wg"])                     # ... also after the captured steps (lock-step)
    assert torch.isfinite(got[0]["logits"]).all()


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--shard-optimizer"], ["--reduce-dtype", "bf16"]])
def test_rccl_runs_the_reducers_call_pattern_at_world_size_one(dev, extra):
    """No multi-GPU box exists in this pipeline, so RCCL itself had never executed this code.  A ONE-rank RCCL group with
    VMR_DP_FORCE_COLLECTIVES=1 makes the reducers issue every collective although each is an identity: async all-reduce
    (ncclAvg) of the stage ranges between the replays of the piecewise hipGraphs, reduce-scatter + scalar all-reduce +
    all-gather for the sharded optimizer, the bf16 staging buffer.  The run must finish, report the nccl backend and a
    finite loss (bench.py asserts it), at the real cfg2 sizes."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, VMR_DP_FORCE_COLLECTIVES="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", RANK="0", WORLD_SIZE="1",
               LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-split", "--steps", "4", "--warmup", "3",
                        "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["dist"] is not None and out["dist"]["backend"] == "nccl" and out["dist"]["world_observed"] == 1
    assert out["step_form"].endswith("piecewise graphs (multi-rank form)") and np.isfinite(out["final_loss"])

