import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
from tests.helpers import load_golden
from tests.test_gpu_trainer import build, arena_grads, graphed
dev = torch.device("cuda")

def run(variant):
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
    first = eager(); eager()
    named = dict(B.named_parameters())
    res = []
    for it in range(3):
        for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v), (optA.step_t, optB.step_t)):
            dst.copy_(src)
        optA.sync_mirrors()
        if variant == "sync_before_replay":
            torch.cuda.synchronize()
        before = optB.arena.flat_p.clone()
        la = float(stepA().item())
        lb = eager()
        torch.cuda.synchronize()
        res.append((round(la, 4), round(lb, 4)))
        if variant != "no_tensor_loop":
            gB = arena_grads(optB, B)
            gmax = max(float(v.abs().max()) for v in gB.values())
            w = 0.0
            for n in optB.names:
                if float(gB[n].abs().max()) < 1e-4 * gmax:
                    continue
                o, k = optB.offsets[n], named[n].numel()
                dA = (optA.arena.flat_p[o:o + k] - before[o:o + k]).double()
                dB = (optB.arena.flat_p[o:o + k] - before[o:o + k]).double()
                w = max(w, float((dA - dB).norm() / dB.norm()))
            res.append(("upd", round(w, 5)))
    print(variant, res, flush=True)
for v in ("exact", "sync_before_replay", "no_tensor_loop", "exact"):
    run(v)
