"""Which factor makes the replay after a state sync disagree with eager?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
from tests.helpers import load_golden
from tests.test_gpu_trainer import build
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
dbatch = {k: v.to(dev) for k, v in batch.items()}
sched = dict(warmup_steps=0.0, total_steps=10)

def run(variant):
    Am = build(cfg, weights, "bf16", dev, g)
    optA = FlatAdamW(Am, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
    stepA = GraphedTrainStep(Am, optA, V.train_engine_SeqPAN, cfg, warmup=2).capture(batch)
    Bm = build(cfg, weights, "bf16", dev, g)
    optB = FlatAdamW(Bm, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
    def eager():
        loss, _ = V.train_engine_SeqPAN(Bm, dbatch, cfg, "train")
        optB.zero_grad(); loss.backward(); optB.step()
        return float(loss.item())
    eager(); eager()
    res = []
    for it in range(3):
        if variant != "nosync_state":
            for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v), (optA.step_t, optB.step_t)):
                dst.copy_(src)
            optA.sync_mirrors()
        if variant == "devsync":
            torch.cuda.synchronize()
        if variant == "eagerA":
            with torch.no_grad():
                V.train_engine_SeqPAN(Am, dbatch, cfg, "train")
        before = optB.arena.flat_p.clone()
        la = float(stepA().item())
        lb = eager()
        res.append((round(la, 4), round(lb, 4)))
    print(variant, res, flush=True)

for v in ("plain", "devsync", "eagerA", "nosync_state", "plain"):
    run(v)
