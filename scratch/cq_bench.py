"""Stand-alone timing of the fused CQAttention core at cfg2 shapes (both directions), graph-free.
usage: cq_bench.py [iters]"""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import ops
dev = torch.device("cuda")
it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B, T, Lq, D = 64, 128, 20, 1024
torch.manual_seed(0)
dt = torch.bfloat16
def mk(*s): return (torch.randn(*s, device=dev) / 4).to(dt).requires_grad_(True)
V, Q = mk(B, T, D), mk(B, Lq, D)
bop, aop = mk(B, Lq, D), mk(B, Lq, D)
colterm = torch.randn(B, Lq, device=dev, requires_grad=True)
vm = torch.ones(B, T, device=dev); qm = torch.ones(B, Lq, device=dev)
g1 = torch.randn(B * T, 4 * D, device=dev).to(dt); g2 = torch.randn(B * Lq, 4 * D, device=dev).to(dt)
def step():
    o1 = ops.cq_block(V, Q, V, bop, colterm, vm, qm, 0)          # q2v: context = video
    o2 = ops.cq_block(Q, V, V, aop, colterm, vm, qm, 1)          # v2q: context = query
    torch.autograd.backward([o1, o2], [g1, g2])
for _ in range(3): step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(it): step()
e1.record(); torch.cuda.synchronize()
print(f"cq block fwd+bwd both directions: {e0.elapsed_time(e1) / it * 1e3:.1f} us per step-equivalent")
