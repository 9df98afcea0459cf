import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
wl = sys.argv[1] if len(sys.argv) > 1 else "basefast"
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
a = dict(Bn.CFG4 if os.environ.get("BF_CFG", "4" if wl == "basefast" else "2") == "4" else Bn.CFG2)
if "BF_DROP" in os.environ: a["droprate"] = float(os.environ["BF_DROP"])
Model, engine = (V.BaseFast, V.train_engine_BaseFast) if wl == "basefast" else (V.SeqPAN, V.train_engine_SeqPAN)
torch.manual_seed(1234)
cfg = Bn.make_cfg(a, os.environ.get("BF_DTYPE", "bf16")); cfg.device = dev
glove = np.random.default_rng(1234).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
model = Model(cfg, glove).to(dev); model.sync_timing = False; model.base_seed = 1234
total = 210
opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=10 * total)
batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
model.train()
graph = len(sys.argv) <= 2
if graph:
    step = GraphedTrainStep(model, opt, engine, cfg, None, warmup=3).capture(batch)
else:
    def step():
        loss, out = engine(model, batch, cfg, "train")
        opt.zero_grad(); loss.backward(); opt.step()
        return loss
for it in range(total):
    loss = step()
    if it % 10 == 0 or it >= 28 or not np.isfinite(float(loss.item())):
        gn = float(opt.gnorm_sq.item()) ** 0.5
        pmax = float(opt.arena.flat_p.abs().max())
        extra = ""
        if graph:
            o = step.out
            extra = " logits_finite %s" % bool(torch.isfinite(o["slogits"].float()).all() and torch.isfinite(o["elogits"].float()).all())
        print(it, float(loss.item()), "gnorm", gn, "pmax", pmax, "step_t", int(opt.step_t.item()), extra, flush=True)
        if it >= 40: break
        if not np.isfinite(float(loss.item())): break
