"""torch.profiler over ONE eager train step: which Python lines launch the remaining torch glue (aten ops with device time)."""
import collections, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
a = Bn.CFG2; cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
glove = np.random.default_rng(1234).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
torch.manual_seed(1234)
model = V.SeqPAN(cfg, glove).to(dev); model.sync_timing = False
opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=100)
batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
model.train()
def step():
    loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
agg = collections.defaultdict(lambda: [0, 0.0])
for e in prof.events():
    dt = getattr(e, "device_time_total", 0) or getattr(e, "cuda_time_total", 0)
    if not e.name.startswith("aten::") or dt <= 0 or e.cpu_children and any(c.name.startswith("aten::") and (getattr(c, "device_time_total", 0) or 0) > 0 for c in e.cpu_children):
        continue
    fr = [s for s in (e.stack or []) if "vmrframe_amd" in s or "bench.py" in s or "glue_profile" in s]
    where = fr[0].split("/")[-1] if fr else ((e.stack or ["<autograd engine>"])[0][-60:])
    k = (e.name, str(e.input_shapes)[:70], where[:70])
    agg[k][0] += 1; agg[k][1] += dt
tot = 0
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += t
    print(f"{t:8.1f} us {n:3d}x {k[0]:28s} {k[1]:70s} {k[2]}")
print("total aten device time per step: %.1f us" % tot)
