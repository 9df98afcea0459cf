"""Debug: BAN eager vs eager vs graphed step trajectories with FlatAdamW at anet layer sizes (B = 2), fixed proposals."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.gen_golden_ban_anet import make_cfg, make_inputs, recipe_weights
from vmrframe_amd.ban import BAN, train_engine_BAN
from vmrframe_amd.ban_trainer import GraphedBANStep
from vmrframe_amd.optim import FlatAdamW
dev = torch.device("cuda")
dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
cfg = make_cfg(dev)
data = {k: torch.from_numpy(v).to(dev) for k, v in make_inputs().items()}
g2 = np.load("tests/golden/g_ban_anet.npz")
fixed = torch.from_numpy(g2["out_coarse_pred"].reshape(2, 80, 2))
def build():
    torch.manual_seed(5)
    m = BAN(cfg, pre_train_emb=np.random.default_rng(1).standard_normal((200, 300)).astype(np.float32), compute_dtype=dtype, sync_timing=False).to(dev).eval()
    W = recipe_weights({k: tuple(p.shape) for k, p in m.named_parameters() if "glove" not in k and "pad_vec" not in k})
    m.load_state_dict({k: torch.from_numpy(v) for k, v in W.items()}, strict=False)
    m.sample = lambda cells: fixed
    return m
def eager(n, clear=False):
    m = build(); o = FlatAdamW(m, lr=1e-4, max_norm=1.0, loss_scale=1024.0 if dtype == torch.float16 else None)
    out = []
    for _ in range(n):
        o.zero_grad(); l, _ = train_engine_BAN(m, data, cfg, "train"); o.backward(l); o.step()
        if clear:
            for c in m._cache.caches: c.store.clear()
        out.append(float(l.detach()))
    return out, m
ea, ma = eager(5); eb, mb = eager(5); ec, mc = eager(5, clear=True)
print("eager A", ea); print("eager B", eb); print("eager C (caches wiped every step)", ec)
m = build(); o = FlatAdamW(m, lr=1e-4, max_norm=1.0, loss_scale=1024.0 if dtype == torch.float16 else None)
gs = GraphedBANStep(m, o, cfg, warmup=2).capture(data)
print("graph  ", [float(gs().detach()) for _ in range(3)])
print("param diff A-B", max(float((p - q).abs().max()) for p, q in zip(ma.parameters(), mb.parameters())),
      "A-C", max(float((p - q).abs().max()) for p, q in zip(ma.parameters(), mc.parameters())),
      "A-graph", max(float((p - q).abs().max()) for p, q in zip(ma.parameters(), m.parameters())))
