import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
x = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
h = dist.all_reduce(x, op=dist.ReduceOp.AVG, async_op=True); h.wait()
torch.cuda.synchronize()
print("RCCL AVG ok:", float(x[5]), torch.cuda.nccl.version() if hasattr(torch.cuda, "nccl") else "")
dist.barrier(); dist.destroy_process_group()
