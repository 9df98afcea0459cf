import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
a = dict(Bn.CFG2); a["droprate"] = 0.0
torch.manual_seed(1234)
cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
glove = np.random.default_rng(1234).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
model = V.BaseFast(cfg, glove).to(dev); model.sync_timing = False; model.base_seed = 1234
opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=2100)
batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
model.train()
step = GraphedTrainStep(model, opt, V.train_engine_BaseFast, cfg, None, warmup=3).capture(batch)
fin = lambda t: bool(torch.isfinite(t.float()).all())
def state():
    A = opt.arena
    d = {"flat_p": fin(A.flat_p), "flat_w": fin(A.flat_w), "flat_wt": fin(A.flat_wt), "m": fin(opt.m), "v": fin(opt.v), "flat_g": fin(A.flat_g),
         "gmax": float(A.flat_g.abs().max()), "vmax": float(opt.v.max()), "mmax": float(opt.m.abs().max())}
    inside = {id(p) for p in A.params}
    for n, p in model.named_parameters():
        if id(p) not in inside and not fin(p): d["outside:" + n] = False
    for n, p in model.named_buffers():
        if p.is_floating_point() and not fin(p): d["buffer:" + n] = False
    return d
for it in range(40):
    loss = step()
    torch.cuda.synchronize()
    lv = float(loss.item())
    o = step.out
    okl = fin(o["slogits"]) and fin(o["elogits"])
    if it >= 12 or not np.isfinite(lv):
        print(it, lv, "logits ok" if okl else "LOGITS BAD", {k: v for k, v in state().items() if v is False or k in ("gmax", "vmax", "mmax")}, flush=True)
    if not np.isfinite(lv):
        bad = [n for n in opt.names if not fin(opt.arena.flat_g[opt.offsets[n]:opt.offsets[n] + dict(model.named_parameters())[n].numel()])]
        print("nonfinite grads:", bad[:12], len(bad))
        break
