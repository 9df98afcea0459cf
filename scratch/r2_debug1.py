"""debug: forward determinism, load_state_dict refresh, replay after state sync"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
from tests.helpers import load_golden
from tests.test_gpu_trainer import build, arena_grads, rel
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
dbatch = {k: v.to(dev) for k, v in batch.items()}
m = build(cfg, weights, "bf16", dev, g)
outs = []
for i in range(3):
    loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
    outs.append((loss.item(), out["slogits"].detach().float().clone()))
print("fresh model, 3 forwards: loss", [o[0] for o in outs], "max|d slogits|", float((outs[0][1]-outs[1][1]).abs().max()), float((outs[0][1]-outs[2][1]).abs().max()))
opt = FlatAdamW(m, lr=0.0, max_norm=1.0)
gs = []
for i in range(4):
    loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
    opt.zero_grad(); loss.backward(); opt.step()
    gs.append((loss.item(), out["slogits"].detach().float().clone(), opt.arena.flat_g.clone()))
print("lr=0 arena steps: loss", [x[0] for x in gs])
print("  slogits vs fresh:", [float((x[1]-outs[0][1]).abs().max()) for x in gs])
print("  grad rel between consecutive arena steps:", [rel(gs[i][2], gs[i+1][2]) for i in range(1, 3)])
# which tensors differ between arena forward and fresh forward? check the mirror against a cast of the master
A = opt.arena
print("mirror == master.to(bf16):", bool(torch.equal(A.flat_w, A.flat_p.to(torch.bfloat16))))
# replay after state sync
sched = dict(warmup_steps=0.0, total_steps=10)
Am = build(cfg, weights, "bf16", dev, g)
optA = FlatAdamW(Am, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
stepA = GraphedTrainStep(Am, optA, V.train_engine_SeqPAN, cfg, warmup=2).capture(batch)
Bm = build(cfg, weights, "bf16", dev, g)
optB = FlatAdamW(Bm, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
def eager(mm, oo):
    loss, _ = V.train_engine_SeqPAN(mm, dbatch, cfg, "train")
    oo.zero_grad(); loss.backward(); oo.step()
    return float(loss.item())
eager(Bm, optB); eager(Bm, optB)
for it in range(3):
    for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v), (optA.step_t, optB.step_t)):
        dst.copy_(src)
    optA.sync_mirrors()
    torch.cuda.synchronize()
    print(it, "flat_w equal:", bool(torch.equal(optA.arena.flat_w, optB.arena.flat_w)), "flat_wt equal:", bool(torch.equal(optA.arena.flat_wt, optB.arena.flat_wt)),
          "mirror==cast:", bool(torch.equal(optA.arena.flat_w, optA.arena.flat_p.to(torch.bfloat16))))
    with torch.no_grad():
        la_e, _ = V.train_engine_SeqPAN(Am, dbatch, cfg, "train")
        lb_e, _ = V.train_engine_SeqPAN(Bm, dbatch, cfg, "train")
    la = float(stepA().item())
    lb = eager(Bm, optB)
    torch.cuda.synchronize()
    print(it, "eager fwd A", float(la_e), "eager fwd B", float(lb_e), "replay A", la, "eager step B", lb, "step_t", int(optA.step_t), int(optB.step_t))
