import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from tests.helpers import load_golden
from tests.test_gpu_trainer import build, arena_grads, graphed
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
sched = dict(warmup_steps=0.0, total_steps=10)
A = build(cfg, weights, "bf16", dev, g)
optA, stepA = graphed(A, cfg, batch, 1e-3, dev, **sched)
B = build(cfg, weights, "bf16", dev, g)
optB = FlatAdamW(B, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
dbatch = {k: v.to(dev) for k, v in batch.items()}
def eager():
    loss, _ = V.train_engine_SeqPAN(B, dbatch, cfg, "train")
    optB.zero_grad(); loss.backward(); optB.step()
    return float(loss.item())
eager(); eager()
named = dict(B.named_parameters())
for it in range(3):
    for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v), (optA.step_t, optB.step_t)):
        dst.copy_(src)
    optA.sync_mirrors()
    before = optB.arena.flat_p.clone()
    la = float(stepA().item())
    gA = optA.arena.flat_g.clone()
    nA = float(optA.gnorm_sq.item())
    lb = eager()
    torch.cuda.synchronize()
    gBf = optB.arena.flat_g
    print(it, "la", la, "lb", lb, "gnorm_sq", nA, float(optB.gnorm_sq.item()), "grad rel diff", float((gA - gBf).norm() / gBf.norm()))
    rows = []
    for n in optB.names:
        o, k = optB.offsets[n], named[n].numel()
        ga, gb = gA[o:o + k], gBf[o:o + k]
        dA = (optA.arena.flat_p[o:o + k] - before[o:o + k]).double()
        dB = (optB.arena.flat_p[o:o + k] - before[o:o + k]).double()
        rows.append((float((dA - dB).norm() / (dB.norm() + 1e-30)), float((ga - gb).norm() / (gb.norm() + 1e-30)), float(gb.norm()), n))
    rows.sort(reverse=True)
    for r in rows[:6]:
        print("   upd rel %.3e  grad rel %.3e  |gB| %.3e  %s" % r)
