import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.helpers import load_golden
from tests.test_gpu_trainer import build, rel
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
dbatch = {k: v.to(dev) for k, v in batch.items()}
def fwd(m):
    loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
    return loss.item(), out["slogits"].float().detach().clone()
f1 = build(cfg, weights, "bf16", dev, g); f2 = build(cfg, weights, "bf16", dev, g)
a = [fwd(f1) for _ in range(3)]; b = fwd(f2)
print("fresh same-model run-to-run:", [rel(x[1], a[0][1]) for x in a[1:]], "other fresh:", rel(b[1], a[0][1]))
m = build(cfg, weights, "bf16", dev, g)
opt = FlatAdamW(m, lr=1e-2, max_norm=1.0)
for _ in range(2):
    loss, _ = V.train_engine_SeqPAN(m, dbatch, cfg, "train"); opt.zero_grad(); loss.backward(); opt.step()
mv = fwd(m)
m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in weights.items()})
r = [fwd(m) for _ in range(3)]
print("moved vs fresh:", rel(mv[1], a[0][1]), "reloaded vs fresh:", [rel(x[1], a[0][1]) for x in r], "losses", a[0][0], [x[0] for x in r])
# arena model without training: forward noise with the arena path
m2 = build(cfg, weights, "bf16", dev, g); o2 = FlatAdamW(m2, lr=0.0, max_norm=1.0)
for _ in range(2):
    loss, _ = V.train_engine_SeqPAN(m2, dbatch, cfg, "train"); o2.zero_grad(); loss.backward(); o2.step()
r2 = [fwd(m2) for _ in range(3)]
print("arena (lr=0) vs fresh:", [rel(x[1], a[0][1]) for x in r2])
