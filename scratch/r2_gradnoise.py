import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import load_golden
from tests.test_gpu_trainer import build, rel
import vmrframe_amd as V
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
dbatch = {k: v.to(dev) for k, v in batch.items()}
def grads():
    m = build(cfg, weights, "bf16", dev, g)
    loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
    loss.backward()
    return {n: p.grad.detach().float().cpu() for n, p in m.named_parameters() if p.grad is not None}, float(loss.item())
runs = [grads() for _ in range(4)]
tot = sum(float(v.double().pow(2).sum()) for v in runs[0][0].values()) ** 0.5
print("losses", [r[1] for r in runs])
for i in range(1, 4):
    rows = sorted(((rel(runs[i][0][k], runs[0][0][k]), k, float(runs[0][0][k].norm()) / tot) for k in runs[0][0]), reverse=True)
    rows = [r for r in rows if r[2] >= 1e-3]
    print("run", i, "worst (share >= 1e-3):", [(round(a, 4), k, round(s, 4)) for a, k, s in rows[:6]])
    print("    w4C/w4Q:", [(round(a, 4), k) for a, k, s in rows if "w4" in k])
