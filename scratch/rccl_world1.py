"""RCCL at world_size 1 on the one GPU of a gpurun box: the raw collectives dp.GradReducer / dp.ShardedReducer issue
(all_reduce AVG / SUM on fp32 slices, reduce_scatter_tensor, all_gather_into_tensor on a 16-bit buffer).  Prints one
line per check.  The reducers themselves on the real step: VMR_DP_FORCE_COLLECTIVES=1 python bench.py --force-split."""
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
dist.destroy_process_group()
print("done (the reducers themselves: VMR_DP_FORCE_COLLECTIVES=1 python bench.py --force-split [--shard-optimizer])")
