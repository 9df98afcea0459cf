"""lr = 0: every replay must give the same loss.  Which interleaved eager work breaks it?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from tests.helpers import load_golden
from tests.test_gpu_trainer import build, arena_grads, graphed
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
A = build(cfg, weights, "bf16", dev, g)
optA, stepA = graphed(A, cfg, batch, 0.0, dev)
B = build(cfg, weights, "bf16", dev, g)
optB = FlatAdamW(B, lr=0.0, weight_decay=0.01, max_norm=1.0)
dbatch = {k: v.to(dev) for k, v in batch.items()}
def eager():
    loss, _ = V.train_engine_SeqPAN(B, dbatch, cfg, "train")
    optB.zero_grad(); loss.backward(); optB.step()
    return float(loss.item())
eager(); eager()
named = dict(B.named_parameters())
def loop_abs():
    gB = arena_grads(optB, B)
    return max(float(v.abs().max()) for v in gB.values())
def loop_sub():
    w = 0.0
    for n in optB.names:
        o, k = optB.offsets[n], named[n].numel()
        dA = (optA.arena.flat_p[o:o + k] - optB.arena.flat_p[o:o + k]).double()
        w = max(w, float(dA.norm()))
    return w
print("replays alone:", [round(float(stepA().item()), 4) for _ in range(4)], flush=True)
print("replay, eagerB:", [(round(float(stepA().item()), 4), round(eager(), 4)) for _ in range(3)], flush=True)
r = []
for _ in range(3):
    loop_abs(); r.append(round(float(stepA().item()), 4))
print("loop_abs then replay:", r, flush=True)
r = []
for _ in range(3):
    loop_sub(); r.append(round(float(stepA().item()), 4))
print("loop_sub then replay:", r, flush=True)
r = []
for _ in range(3):
    eager(); torch.cuda.synchronize(); loop_abs(); loop_sub(); optA.sync_mirrors(); x = optB.arena.flat_p.clone(); r.append((round(float(stepA().item()), 4), round(eager(), 4)))
print("test-like sequence:", r, flush=True)
print("replays alone again:", [round(float(stepA().item()), 4) for _ in range(4)], flush=True)
torch.cuda.synchronize()
print("mirror intact:", bool(torch.equal(optA.arena.flat_w, optA.arena.flat_p.to(torch.bfloat16))), "A==B masters:", bool(torch.equal(optA.arena.flat_p, optB.arena.flat_p)))
