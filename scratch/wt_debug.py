import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tests.test_gpu_model as T
from tests.helpers import load_golden
import vmrframe_amd as V
from vmrframe_amd import ops
from vmrframe_amd.optim import FlatAdamW
dev = torch.device("cuda")
if "--pre" in sys.argv:
    T.test_flat_arena_direct_accumulation_and_fused_adamw(dev)
z, cfg, batch, g, weights = load_golden("g_small")
cfg.device = dev
def fresh():
    m = T.build(cfg, weights, "bf16", dev); m.gumbel_override = g.to(dev); m.eval(); return m
ma, mb = fresh(), fresh()
oa, ob = FlatAdamW(ma, lr=1e-3, max_norm=1.0), FlatAdamW(mb, lr=1e-3, max_norm=1.0)
for it in range(3):
    for m, o, use in ((ma, oa, True), (mb, ob, False)):
        ops.USE_WT = use
        loss, _ = V.train_engine_SeqPAN(m, batch, cfg, "train")
        o.zero_grad(); loss.backward(); o.step()
    if it == 0: continue
    ga, gb = oa.arena.flat_g, ob.arena.flat_g
    print("it", it, "rel", float((ga - gb).norm() / gb.norm()), "norms", float(ga.norm()), float(gb.norm()))
    rows = []
    for (n, p), (_, q) in zip(ma.named_parameters(), mb.named_parameters()):
        if p.grad is None: continue
        d = float((p.grad - q.grad).norm()); rows.append((d, float(q.grad.norm()), n))
    for d, nb, n in sorted(rows, reverse=True)[:8]:
        print(f"   {d:10.4e} of {nb:10.4e}  {n}")
    pa = torch.cat([p.detach().reshape(-1) for p in ma.parameters()]); pb = torch.cat([p.detach().reshape(-1) for p in mb.parameters()])
    print("   weights rel diff", float((pa - pb).norm() / pb.norm()))
