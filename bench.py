#!/usr/bin/env python3
"""bench.py -- SeqPAN train-step throughput on MI355X (BASELINE.json metric:
clips/sec of a train step at B x T = 64 x 128, D = 1024, bf16).

One "step" = the reference loop body main.py:88-97 on one synthetic batch that is
already resident in HBM: forward, both losses, backward, gradient all-reduce
(N > 1), clip_grad_norm_(1.0), AdamW (two decay groups), linear-warmup schedule.
Dropout is ON (droprate 0.2, the reference's configs); weights are random-init.

  python bench.py --gpus N --steps K --warmup W
For N > 1 it runs one rank per GPU over RCCL: either under torch.distributed.run (RANK / WORLD_SIZE /
MASTER_* in the environment) or, when those are absent, by spawning its own N rank processes before anything
touches a GPU.  Per-GPU batch is fixed at 64 clips (weak scaling).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CFG2 = dict(B=64, T=128, L=20, D=1024, V=500, num_words=4002, num_chars=60, C=8, droprate=0.2)
# BASELINE configs[3] ("next" row N1): BaseFast path, T=256, D=1024, V=1024 -- NOT the headline metric
CFG4 = dict(B=64, T=256, L=20, D=1024, V=1024, num_words=4002, num_chars=60, C=8, droprate=0.2)
# BASELINE configs[4] ("next" row N2, first slice): the BAN 2-D proposal-map stage alone -- NOT the headline metric
CFG5 = dict(B=64, N=128, F=512, Cd=128, pooling=[31, 16, 16], min_iou=0.5, max_iou=1.0)
TRAIN_GFLOP_PER_CLIP = 58.0      # 19.34 GFLOP fwd (FlopCounter on the reference) x 3
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak


def run_ban(args, dev, rank, world):
    """BASELINE.json configs[4] as a whole: the BAN train step (reference models/BAN.py: forward incl. the host-side proposal
    sampler, the five losses of train_engine_BAN, backward, AdamW) at config/anet/BAN.yaml's model sizes with T = 128 -> a
    128 x 128 score map (5376 kept cells, 80 proposals per clip), B = 64 clips per GPU; --dtype fp16 is configs[4]'s dtype
    (dynamic loss scale carried on the device by FlatAdamW), bf16 / fp32 the others.  The optimizer is the flat fused AdamW
    (VMR_BAN_TORCH_ADAMW=1: torch.optim.AdamW, no loss scale -> bf16 / fp32 only).  N > 1 = independent replicas (no gradient
    exchange is wired for this row yet): "replicas only"."""
    from types import SimpleNamespace as NS
    import vmrframe_amd as V
    torch.manual_seed(1234 + rank)
    B, T, Vw, E, Lq = CFG5["B"], CFG5["N"], 4000, 300, 20
    cfg = NS(device=dev,
             model=NS(vlen=T, topk=20, neighbor=3, negative=0, prop_num=80, sparse_sample=True, pooling_counts=CFG5["pooling"],
                      fuse_dim=CFG5["F"], vdim=1024, dim=CFG5["F"] // 2, lstm_layer=2, query_embed_dim=E, contrast_dim=CFG5["Cd"],
                      droprate=0.1, gcn=NS(num_blocks=2, k=80, hidden_size=CFG5["F"])),
             loss=NS(min_iou=CFG5["min_iou"], max_iou=CFG5["max_iou"], bce=2.0, refine=1.0, td=0.1, offset=1.0, contrast=0.1))
    rng = np.random.default_rng(1234)
    model = V.BAN(cfg, pre_train_emb=rng.standard_normal((Vw, E)).astype(np.float32),
                  compute_dtype={"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.dtype],
                  sync_timing=False).to(dev).train()
    torch_opt = os.environ.get("VMR_BAN_TORCH_ADAMW", "0") == "1"
    assert not (torch_opt and args.dtype == "fp16"), "fp16 needs the loss scale FlatAdamW carries"
    from vmrframe_amd.optim import FlatAdamW
    if torch_opt:
        opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01)
    else:
        opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0)
    gen = torch.Generator().manual_seed(1234 + rank)
    vl = torch.randint(T // 2, T + 1, (B,), generator=gen); vl[0] = T
    ql = torch.randint(5, Lq + 1, (B,), generator=gen); ql[0] = Lq
    data = {"vfeats": torch.randn(B, T, 1024, generator=gen), "words_ids": torch.randint(1, Vw + 2, (B, Lq), generator=gen),
            "vlens": vl, "tlens": ql, "start_end_offset": torch.randn(B, T, T, 2, generator=gen),
            "iou2ds": torch.rand(B, T, T, generator=gen), "dist_idxs": torch.rand(B, 2, T, generator=gen),
            "map2d_contrasts": torch.rand(B, 2, T, T, generator=gen) > 0.5}
    data = {k: v.to(dev) for k, v in data.items()}

    def step():
        if torch_opt:
            opt.zero_grad(set_to_none=True)
        else:
            opt.zero_grad()
        loss, _ = V.train_engine_BAN(model, data, cfg, "train")
        if torch_opt:
            loss.backward()
        else:
            opt.backward(loss)
        opt.step()
        return loss

    graphed = not args.no_graph
    if graphed:      # two hipGraphs around the sampler's host round trip (vmrframe_amd/ban_trainer.py)
        from vmrframe_amd.ban_trainer import GraphedBANStep
        if torch_opt:
            opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=0.01,
                                    fused=True, capturable=True)
        step = GraphedBANStep(model, opt, cfg, warmup=3).capture(data)
    for _ in range(max(3, args.warmup)):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # outside the timed region: no workgroup of a one-launch recurrence ran into its poll bound (a result computed after
    # that is garbage: fail loudly), and the loss is a number
    from vmrframe_amd import ban_encoders as _enc
    assert not _enc.seq_kernel_gave_up(), "a one-launch LSTM recurrence gave up waiting for its exchange (error word raised)"
    assert np.isfinite(float(loss.detach())), "BAN training diverged"
    if rank == 0:
        ms = dt / args.steps * 1e3
        print(json.dumps({"metric": "clips/sec (train step), BAN at T=128 (128x128 score map), anet model sizes",
                          "value": round(B * world / (ms * 1e-3), 2), "unit": "clips/sec", "n_gpus": world, "steps": args.steps,
                          "warmup": max(3, args.warmup), "ms_per_step": round(ms, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": "BAN (configs[4], next-row N2): B=64 clips/GPU, T=128, 20-word queries, vdim 1024, dim 256, "
                                                 "2 LSTM layers, fuse_dim 512, 5376 map cells, 80 proposals; forward (proposal sampler on the device "
                                                 "included) + five losses + backward + clip + AdamW (" +
                                                 ("torch.optim" if torch_opt else "flat fused, loss scale %g" % opt.loss_scale()) +
                                                 "); one hipGraph (VMR_BAN_DEVICE_SAMPLER=0: two, around the host sampler) unless --no-graph; replicas only for N > 1",
                                     "global_batch": B * world, "parallelism": f"dp{world}"},
                          "final_loss": round(float(loss.detach()), 4), "hipgraph": graphed, "roofline": None, "cpu_baseline": None}))
    return 0


def make_cfg(a, dtype):
    from vmrframe_amd import synth as S    # config object + synthetic batch recipe (no oracle on the product path)
    cfg = S.make_cfg(dim=a["D"], vlen=a["T"], vdim=a["V"], num_words=a["num_words"], num_chars=a["num_chars"],
                     droprate=a["droprate"])
    cfg.model.compute_dtype = dtype
    return cfg


def synth(a, seed):
    from vmrframe_amd import synth as S
    return S.synth_batch(a["B"], a["T"], a["L"], a["V"], a["num_words"], a["num_chars"], C=a["C"], seed=seed)


# A timed launch is issued REPS times back to back between ONE event pair: in the eager instrumented steps the GPU sits
# idle before every launch and an event pair around a single 30-70 us kernel reads 8-15 us long (round 2: 78 us between
# events against 63 us in the rocprofv3 trace of the replayed graph); back to back the launch gaps are those of the
# graph replay.  The products are pure functions of their operands; the merged launches additionally ACCUMULATE (the
# ridden slab reduction, the bias column sums), so the gradients of the two instrumented steps -- taken after the timed
# region and after the reported loss -- are over-counted, which nothing reads.
TIMER_REPS = 3


class GemmTimer:
    """HIP events around every launch of the forward x.W^T kernel (the NT bf16 MFMA GEMM) on the
    stream it is launched on; achieved TFLOP/s = sum(flops) / sum(event time)."""

    def __init__(self):
        self.records = []

    def __call__(self, launch, M, N, K, ta, tb, Z, dtype):
        # only the launches the library routes to the LDS-DMA kernel (gemm.hip: M, N multiples of 128,
        # K a multiple of 64, unbatched) -- the same population the rocprof / PMC summaries aggregate
        if ta or tb or dtype != 1 or Z != 1 or M % 128 or N % 128 or K % 64 or K < 128:
            return launch()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(TIMER_REPS):
            launch()
        e.record()
        self.records.append((s, e, 2.0 * M * N * K * Z, 2.0 * (M * K + N * K + M * N)))

    def summary(self):
        if not self.records:
            return None
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records) / TIMER_REPS
        fl = sum(r[2] for r in self.records)
        return {"launches": len(self.records), "ms_total": ms, "tflops": fl / (ms * 1e-3) / 1e12,
                "avg_us": ms * 1e3 / len(self.records), "flops_per_launch": fl / len(self.records),
                "bytes_per_launch": sum(r[3] for r in self.records) / len(self.records)}


class Gemm2Timer:
    """HIP events around the merged launches (gemm_bf16_dma2_kernel: a layer's dW + dX products and the previous
    layer's slab reduction in one grid)."""

    def __init__(self):
        self.records = []

    def __call__(self, launch, flops, nbytes):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(TIMER_REPS):
            launch()
        e.record()
        self.records.append((s, e, flops, nbytes))

    def summary(self):
        if not self.records:
            return None
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records) / TIMER_REPS
        fl = sum(r[2] for r in self.records)
        return {"launches": len(self.records), "ms_total": ms, "tflops": fl / (ms * 1e-3) / 1e12,
                "avg_us": ms * 1e3 / len(self.records), "flops_per_launch": fl / len(self.records),
                "bytes_per_launch": sum(r[3] for r in self.records) / len(self.records)}


def cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(a, B=16, n=3):
    """The oracle (CPU fp32 restatement of the reference, kind 'port') timed on the host cores on a BOUNDED sample of
    the same workload: cfg2 shapes at B = 16 clips, one warm-up + three timed train steps (fwd + losses + bwd + clip +
    AdamW, dropout on).  B = 64 itself takes 44 s per step on 8 cores of the build container (1.44 clips/s; the
    reference's own code: 45.0 s) -- beyond the few minutes the default bench may take -- and the CPU
    step is memory-bound, so smaller batches read HIGHER clips/s (2.2 at B = 8, 2.0 at B = 32, 1.4 at B = 64 there):
    this is a stated baseline, not a target."""
    from oracle import seqpan_ref as R
    # the GPU box gives one job a 16-CPU share whatever os.cpu_count() says: oversubscribing stalls
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(16, cores))
    torch.set_num_threads(cores)
    b = dict(a); b["B"] = B
    cfg = make_cfg(b, "fp32")
    weights = R.make_weights(cfg, 5)
    P = R.to_params(weights, requires_grad=True)
    batch = synth(b, 5)
    g = R.gumbel_noise(b["B"], b["T"], 5)
    params = [p for p in P.values() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-4)

    def drop(site, x):
        return torch.nn.functional.dropout(x, a["droprate"], True)

    def step():
        loss, _, _ = R.train_loss(P, cfg, batch, g, drop)
        opt.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_([p for p in params if p.grad is not None], 1.0)
        opt.step()
    step()
    print("[bench] cpu_baseline warm-up done", file=sys.stderr, flush=True)
    t0 = time.time()
    for _ in range(n):
        step()
    dt = (time.time() - t0) / n
    return {"value": b["B"] / dt, "unit": "clips/sec", "cores": cores, "kind": "port", "cpu": cpu_model_name(),
            "sample": f"oracle/seqpan_ref.py fp32 train step (fwd+losses+bwd+clip+AdamW, dropout on) at cfg2 "
                      f"shapes with B={B} clips, mean of {n} step(s) after 1 warm-up ({dt:.2f} s/step) on {cores} threads of "
                      f"{cpu_model_name()}"}


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start N copies of this script, one per GPU, with the
    torch.distributed environment (rendezvous on 127.0.0.1), and pass rank 0's JSON line through.  The parent never
    initialises a GPU (no HIP call before or after the spawn) and never replaces itself with another program."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver (RCCL needs it)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        for p in procs:
            rc = p.wait() or rc
            if rc:
                break
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def launcher_selftest(rank: int, world: int):
    """--selftest-launcher: what a rank does up to and including its first collective, without any GPU work (gloo):
    lets the launch path be rehearsed on a GPU-less host (tests/test_dp_gloo.py)."""
    import torch.distributed as dist
    from vmrframe_amd import dp
    dp.init_process_group_from_env("gloo")
    t = torch.ones(4) * (rank + 1)
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launcher_selftest": True, "world_observed": dist.get_world_size(),
                          "sum_of_ranks": float(t[0])}), flush=True)
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # ~1.8 s of timed region at cfg2
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"],
                    help="compute dtype (fp32 masters either way); fp16 = BASELINE configs[4]'s dtype, wired for --workload ban / banmap")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true",
                    help="also time the CPU oracle at the metric's own B = 64 (1 warm-up + 1 step: about a minute on 16 cores)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of whole-step hipGraph replay")
    ap.add_argument("--workload", default="seqpan", choices=["seqpan", "basefast", "banmap", "ban"],
                    help="seqpan = BASELINE configs[1] (headline); basefast = configs[3] (T=256); banmap = the BAN "
                         "proposal-map stage of configs[4]; ban = the whole BAN train step of configs[4] (eager)")
    ap.add_argument("--selftest-launcher", action="store_true", help="rendezvous + one gloo all-reduce per rank, no GPU")
    ap.add_argument("--shard-optimizer", action="store_true",
                    help="reduce-scatter + AdamW on the local 1/N slices + all-gather of the 16-bit mirrors (dp.ShardedReducer) "
                         "instead of all-reduce + replicated AdamW")
    ap.add_argument("--timer-reps", type=int, default=3,
                    help="back-to-back repeats of every timed GEMM launch in the two instrumented steps behind the timed region "
                         "(1 under a profiler, so that the trace's per-step launch counts stay those of the step)")
    ap.add_argument("--force-split", action="store_true",
                    help="at --gpus 1: replay the multi-rank form of the step (one graph per backward stage + the optimizer "
                         "graph(s), reducer hand-over points on the host in between) so its host-hop cost is measured")
    ap.add_argument("--reduce-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="wire dtype of the gradient all-reduce (N > 1); fp32 = exact sum")
    args = ap.parse_args()
    global TIMER_REPS
    TIMER_REPS = max(1, args.timer_reps)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))            # nothing above this line has touched a GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if args.selftest_launcher:
        return launcher_selftest(rank, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback in the product path)"
    # rehearsal hook for a 1-GPU box: VMR_FORCE_DEVICE=0 + VMR_DIST_BACKEND=gloo runs N ranks on one card
    local = int(os.environ.get("VMR_FORCE_DEVICE", local))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import torch.distributed as dist
    import vmrframe_amd as V
    from vmrframe_amd import dp, ops
    from vmrframe_amd.optim import FlatAdamW
    # rehearsal hook for a 1-GPU box: VMR_DP_FORCE_COLLECTIVES=1 --force-split initialises a ONE-rank RCCL group and the
    # reducers issue every collective although each is an identity: RCCL itself then runs this code's call pattern
    # (async all-reduce / reduce-scatter / all-gather between the replays of the piecewise graphs)
    rehearse = world == 1 and os.environ.get("VMR_DP_FORCE_COLLECTIVES", "0") == "1"
    if world > 1 or rehearse:
        dp.init_process_group_from_env(os.environ.get("VMR_DIST_BACKEND", "nccl"))

    if args.workload == "ban":
        return run_ban(args, dev, rank, world)
    if args.workload == "banmap":
        from vmrframe_amd.synth import Cfg
        a = CFG5
        engine = V.train_engine_ProposalMap2D
        torch.manual_seed(1234)
        cfg = Cfg(device=dev, dense_outputs=True, loss=Cfg(min_iou=a["min_iou"], max_iou=a["max_iou"]))
        model = V.ProposalMap2D(a["F"], a["Cd"], a["N"], a["pooling"]).to(dev)
        model.base_seed = 1234 + rank
        torch.manual_seed(1234 + rank)
    else:
        a = CFG2 if args.workload == "seqpan" else CFG4
        Model, engine = (V.SeqPAN, V.train_engine_SeqPAN) if args.workload == "seqpan" else \
            (V.BaseFast, V.train_engine_BaseFast)
        torch.manual_seed(1234)                      # reference main.py:41
        cfg = make_cfg(a, args.dtype)
        cfg.device = dev
        rng = np.random.default_rng(1234)
        glove = rng.standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
        model = Model(cfg, glove).to(dev)
        model.sync_timing = False                    # the reference's in-forward wall-clock syncs are instrumentation
        model.base_seed = 1234 + rank                # per-rank dropout / Gumbel streams
        torch.manual_seed(1234 + rank)
    dp.broadcast_parameters(model)
    if world > 1 and hasattr(model, "backward_plan"):
        model.backward_cuts = True       # cut the backward at the stage boundaries: all-reduce under the next stage
    total_steps = args.steps + args.warmup
    opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0 * total_steps,
                    total_steps=10 * total_steps)
    if args.shard_optimizer:
        reducer = dp.ShardedReducer(model, opt)
    else:
        reducer = dp.GradReducer(model, opt, reduce_dtype=torch.bfloat16 if args.reduce_dtype == "bf16" else torch.float32)
    split = world > 1 or args.force_split
    if split and hasattr(model, "backward_plan"):
        model.backward_cuts = True
    if args.workload == "banmap":
        gen = torch.Generator().manual_seed(1234 + rank)
        batch = {"hidden_b": torch.relu(torch.randn(a["B"], a["N"], a["F"], generator=gen)).to(dev),   # post-ReLU features
                 "fuse_feature": torch.tanh(torch.randn(a["B"], a["N"], a["F"], generator=gen)).to(dev),  # LSTM-range
                 "iou2ds": torch.rand(a["B"], a["N"], a["N"], generator=gen).to(dev)}
    else:
        batch = {k: v.to(dev) for k, v in synth(a, 1234 + rank).items()}   # weak scaling: 64 clips per GPU
    model.train()

    def eager_step():
        loss, out = engine(model, batch, cfg, "train")
        opt.zero_grad()
        reducer.backward(loss)       # loss.backward(), stage by stage when N > 1 (each stage's range goes to RCCL)
        reducer.finish()
        opt.step()
        return loss

    if args.no_graph:
        step = eager_step
        for _ in range(args.warmup):
            loss = step()
    else:
        # one captured HIP graph per step (vmrframe_amd/trainer.py): the eager loop is launch-bound
        from vmrframe_amd.trainer import GraphedTrainStep
        gstep = GraphedTrainStep(model, opt, engine, cfg, reducer if split else None, warmup=3,
                                 force_split=args.force_split).capture(batch)
        step = gstep
        for _ in range(args.warmup):
            loss = step()
    torch.cuda.synchronize()
    if rank == 0:
        print(f"[bench] warm-up done, loss {float(loss.item()):.4f}", file=sys.stderr, flush=True)
    if world > 1 or rehearse:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1 or rehearse:
        dist.barrier()
    dt = time.perf_counter() - t0
    final_loss = float(loss.item())
    # per-launch timing of the dominant kernel: HIP events on the launch stream around every NT bf16
    # GEMM of two more (eager) steps of the same workload -- events cannot be recorded inside a graph replay
    timer = GemmTimer() if rank == 0 else None
    ops.GEMM_HOOK = timer
    timer2 = Gemm2Timer() if rank == 0 else None
    ops.GEMM2_HOOK = timer2
    cq_rec = []

    CQ_REPS = 4     # the score kernel is a pure function of its inputs: launched 4x back to back between the two events so
                    # that the ~8 us an event pair adds around ONE short launch in eager mode does not pass for kernel time

    def cq_hook(launch, B_, Ll, Ls, D_):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record()
        for _ in range(CQ_REPS):
            launch()
        e_.record()
        cq_rec.append((s_, e_, B_, Ll, Ls, D_))
    ops.CQ_HOOK = cq_hook
    cqa_rec = []

    def cqa_hook(launch, which, B_, Lc_, Lq_, D_):
        s_, e_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s_.record(); launch(); e_.record()
        cqa_rec.append((s_, e_, which, B_, Lc_, Lq_, D_))
    ops.CQ_APPLY_HOOK = cqa_hook
    for _ in range(2):          # every rank takes part (the steps contain the gradient all-reduce)
        eager_step()
    torch.cuda.synchronize()
    ops.GEMM_HOOK = None
    ops.GEMM2_HOOK = None
    ops.CQ_HOOK = None
    ops.CQ_APPLY_HOOK = None
    timed_steps_for_hook = 2
    if world > 1 or rehearse:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert np.isfinite(final_loss), "training diverged"

    if rank == 0:
        print(f"[bench] timed region done: {dt / args.steps * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
        clips = a["B"] * world * args.steps
        value = clips / dt
        gs = timer.summary()
        traffic = traffic2 = None
        tpath = os.path.join(ROOT, "profiles", "gemm_traffic.json")
        if os.path.exists(tpath) and args.workload == "seqpan":   # the PMC passes were taken on the headline workload
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("hbm_bytes_per_launch")
                traffic2 = (tj.get("merged") or {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = traffic2 = None
        roofline = None
        if gs:
            roofline = {"kernel": "gemm_bf16_dma_kernel<false,false,64,2,{4|5}> (NT: x.W^T pointwise-conv / projection GEMM)",
                        "bound": "mfma",
                        "achieved": round(gs["tflops"], 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(gs["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic,
                        "launches_per_step": gs["launches"] / timed_steps_for_hook, "avg_launch_us": round(gs["avg_us"], 2),
                        "flops_per_launch": gs["flops_per_launch"],
                        "algorithmic_bytes_per_launch": gs["bytes_per_launch"]}   # A + W + C in bf16 (no epilogue operands)
        metric = {"seqpan": "clips/sec (train step) at BxT=64x128, D=1024",
                  "basefast": "clips/sec (train step), BaseFast at BxT=64x256, D=1024",
                  "banmap": "clips/sec (train step), BAN proposal-map stage at N=128, F=512"}[args.workload]
        wl = {"seqpan": ("SeqPAN anet/C3D synthetic features (configs[1]): B=64 clips/GPU, T=128, "
                         "L=20, D=1024, V=500, droprate 0.2; full train step "
                         "(fwd+losses+bwd+allreduce+clip+AdamW+schedule)"),
              "basefast": ("BaseFast path (configs[3], next-row N1): B=64 clips/GPU, T=256, L=20, D=1024, "
                           "V=1024, droprate 0.2; full train step"),
              "banmap": ("BAN 2-D proposal-map stage (configs[4], next-row N2 first slice; models/BAN.py:87-99 + loss_bce): "
                         "B=64 clips/GPU, N=128, fuse_dim 512, contrast 128, pooling_counts [31,16,16], bf16, dropout 0.1, "
                         "dense tmap / map2d_proj outputs; fwd+loss+bwd+clip+AdamW")}[args.workload]
        g2 = timer2.summary()
        roofline2 = None
        if g2:
            roofline2 = {"kernel": "gemm_bf16_dma2_kernel<{4|5}> (one launch = a layer's dW = dz^T.x split-K slabs + its dX = dz.Wt^T "
                                   "+ the previous layer's slab reduction)",
                         "bound": "mfma", "achieved": round(g2["tflops"], 2), "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(g2["tflops"] / MFMA_BF16_PEAK_TFLOPS, 4), "traffic": traffic2,
                         "launches_per_step": g2["launches"] / timed_steps_for_hook, "avg_launch_us": round(g2["avg_us"], 2),
                         "flops_per_launch": g2["flops_per_launch"], "algorithmic_bytes_per_launch": g2["bytes_per_launch"]}
            # `roofline` = the kernel with the larger share of the step; the other one is reported beside it
            if roofline is None or g2["ms_total"] > gs["ms_total"]:
                roofline, roofline2 = roofline2, roofline
        out = {"metric": metric, "value": round(value, 2),
               "unit": "clips/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": wl,
                          "global_batch": a["B"] * world, "parallelism": f"dp{world}"},
               "step_mfma_frac": round(value * TRAIN_GFLOP_PER_CLIP * 1e9 / world / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4),
               "final_loss": round(final_loss, 4), "hipgraph": not args.no_graph,
               "step_form": ("sharded-optimizer " if args.shard_optimizer else "") +
                            ("piecewise graphs (multi-rank form)" if split and not args.no_graph else "one graph"),
               "dist": None if (world == 1 and not dist.is_initialized()) else {"backend": dist.get_backend(), "world_observed": dist.get_world_size(),
                                                "reduce_dtype": args.reduce_dtype,
                                                "overlap": ("stage-cut backward; per stage: reduce-scatter of the matrix region + "
                                                            "all-reduce of the fp32 region; AdamW on the local slices; all-gather "
                                                            "of the 16-bit mirrors") if args.shard_optimizer else
                                                           "stage-cut backward, one all-reduce per stage range"},
               "roofline": roofline, "roofline_second": roofline2}
        if cq_rec:
            # the CQAttention score kernel (north-star "attention score/softmax/context-gather"): algorithmic
            # bytes = read long + short operand, write both probability matrices (bf16); flops = the QK^T contraction
            us = [s_.elapsed_time(e_) * 1e3 / CQ_REPS for (s_, e_, *_r) in cq_rec]
            _, _, B_, Ll, Ls, D_ = cq_rec[0]
            by = B_ * ((Ll + Ls) * D_ * 2 + 2 * Ll * ((Ls + 7) // 8 * 8) * 4)     # (fp32 probability pair)
            fl = 2.0 * B_ * Ll * Ls * D_
            t_ = sum(us) / len(us) * 1e-6
            out["cq_score_kernel"] = {"kernel": "cq_score_kernel (trilinear QK^T + both masked softmaxes, one launch)",
                                      "avg_launch_us": round(sum(us) / len(us), 2), "launches_per_step": len(us) / timed_steps_for_hook,
                                      "bound": "hbm", "achieved": round(by / t_ / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                      "frac": round(by / t_ / 8e12, 4), "algorithmic_bytes_per_launch": by,
                                      "mfma_frac": round(fl / t_ / (MFMA_BF16_PEAK_TFLOPS * 1e12), 5),
                                      "note": "AI = 16 flop/B: HBM-bound by construction (<= 5 % of MFMA peak stand-alone)"}
        fw = [r for r in cqa_rec if r[2] == "fwd" and r[4] >= r[5]]     # q2v direction (context = video)
        if fw and cq_rec and gs:
            # the north-star's named target, reported under that name: the fused CQ block =
            # score + both softmaxes + apply stage (two fused kernels) + cqa_linear (the LDS-DMA GEMM), q2v direction.
            # FLOPs: SURVEY 8d's 1.12 GFLOP per clip (2*T*4D*D for cqa_linear + the three K = L contractions + the score);
            # bytes: read C, Q once + write the [T, D] output once (the concat stays on chip in the ideal schedule).
            _, _, _, B_, Lc_, Lq_, D_ = fw[0]
            t_apply = sum(r[0].elapsed_time(r[1]) for r in fw) / len(fw) * 1e-3
            t_score = sum(r[0].elapsed_time(r[1]) / CQ_REPS for r in cq_rec if r[3] == Lc_) / max(1, sum(1 for r in cq_rec if r[3] == Lc_)) * 1e-3
            cqa = [r for r in timer.records if abs(r[2] - 2.0 * B_ * Lc_ * 4 * D_ * D_) < 1.0]        # [B*T, D, 4D] products
            t_lin = sum(r[0].elapsed_time(r[1]) / TIMER_REPS for r in cqa) / max(1, len(cqa)) * 1e-3 if cqa else float("nan")
            flops = B_ * (2.0 * Lc_ * 4 * D_ * D_ + 3 * 2.0 * Lc_ * Lq_ * D_ + 2.0 * Lc_ * Lq_ * D_)
            nbytes = B_ * ((Lc_ + Lq_) * D_ * 2 + Lc_ * D_ * 2)
            t_all = t_score + t_apply + t_lin
            out["fused_cq_block"] = {
                "what": "q2v CQAttention block (models/layers.py:417-437): cq_score_kernel + cq_apply_fwd + cqa_linear GEMM",
                "us": {"score": round(t_score * 1e6, 1), "apply": round(t_apply * 1e6, 1), "cqa_linear": round(t_lin * 1e6, 1)},
                "gflop": round(flops / 1e9, 2), "mfma": {"achieved": round(flops / t_all / 1e12, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                                                          "unit": "TFLOP/s", "frac": round(flops / t_all / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4)},
                "hbm": {"achieved": round(nbytes / t_all / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                        "frac": round(nbytes / t_all / 8e12, 4), "algorithmic_bytes": nbytes},
                "apply_kernel_hbm": {"bytes": B_ * ((Lc_ + Lq_) * D_ * 2 + Lc_ * 4 * D_ * 2),
                                     "achieved_GBps": round(B_ * ((Lc_ + Lq_) * D_ * 2 + Lc_ * 4 * D_ * 2) / t_apply / 1e9, 1),
                                     "frac": round(B_ * ((Lc_ + Lq_) * D_ * 2 + Lc_ * 4 * D_ * 2) / t_apply / 8e12, 4),
                                     "note": "stand-alone apply kernel: reads C, Q, writes the [T, 4D] concat"}}
        if args.workload != "seqpan":
            out["step_mfma_frac"] = None     # the 58 GFLOP/clip figure is SeqPAN's
        if world == 1 and not args.no_cpu_baseline and args.workload == "seqpan":
            out["cpu_baseline"] = cpu_baseline(a)
            if args.cpu_baseline_full:     # BASELINE.md's 1.42 clips/s (the reference itself, 8 Xeon cores) is THIS batch size
                out["cpu_baseline_full"] = cpu_baseline(a, B=a["B"], n=1)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
