"""Which weight-gradient slab products still run as their own launch (no dX came along)?  GPU box only."""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd import ops
from vmrframe_amd.optim import FlatAdamW
dev = torch.device("cuda", 0)
a = Bn.CFG2; cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
glove = np.random.default_rng(1234).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
torch.manual_seed(1234)
model = V.SeqPAN(cfg, glove).to(dev); model.sync_timing = False
opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, total_steps=100)
batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
model.train()
def step():
    loss, _ = V.train_engine_SeqPAN(model, batch, cfg, "train"); opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
seen = collections.Counter()
orig = ops.gemm
def spy(A, B, Cm, M, N, K, ta, tb, *args, **kw):
    if ta and tb:
        merged = kw.get("held") is not None and kw.get("splitk", 1) > 1 and (kw.get("flags", 0) & ops.L.EPI_SLAB)
        seen[("merged" if merged else "alone", M, N, K, kw.get("splitk", 1), bool(kw.get("flags", 0) & ops.L.EPI_SLAB))] += 1
    elif kw.get("defer"):
        seen[("dX held", M, N, K, 1, False)] += 1
    return orig(A, B, Cm, M, N, K, ta, tb, *args, **kw)
ops.gemm = spy
step(); torch.cuda.synchronize()
for k, v in sorted(seen.items()): print(v, k)
