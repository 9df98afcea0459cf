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
def segs():
    snap = torch.cuda.memory_snapshot()
    priv = [(s["address"], s["address"] + s["total_size"], s.get("segment_pool_id")) for s in snap if tuple(s.get("segment_pool_id", (0, 0))) != (0, 0)]
    return priv
print("private segments:", len(segs()), "bytes", sum(b - a for a, b, _ in segs()))
for it in range(3):
    for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v), (optA.step_t, optB.step_t)):
        dst.copy_(src)
    optA.sync_mirrors()
    torch.cuda.synchronize()
    with torch.no_grad():
        la_e = float(V.train_engine_SeqPAN(A, dbatch, cfg, "train")[0].item())
    optA.sync_mirrors()     # (drops the eager casts again)
    before = optB.arena.flat_p.clone()
    la = float(stepA().item())
    lb = eager()
    torch.cuda.synchronize()
    print(it, "A eager fwd", round(la_e, 4), "A replay", round(la, 4), "B eager", round(lb, 4), flush=True)
    priv = segs()
    hits = 0
    gB = arena_grads(optB, B)
    temps = []
    for n in optB.names:
        t1 = gB[n].abs(); temps.append((t1.data_ptr(), t1.numel() * t1.element_size()))
        t2 = t1.max(); temps.append((t2.data_ptr(), 4))
        float(t2)
        o, k = optB.offsets[n], named[n].numel()
        dA = (optA.arena.flat_p[o:o + k] - before[o:o + k]).double(); temps.append((dA.data_ptr(), dA.numel() * 8))
    for p, nb in temps:
        for a, b, pid in priv:
            if p < b and p + nb > a:
                hits += 1
    print("   loop temporaries inside a graph-private segment:", hits, "of", len(temps))
