import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW
dev = torch.device("cuda")
a = Bn.CFG2
cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
torch.manual_seed(0)
glove = np.random.default_rng(0).standard_normal((a["num_words"]-2, 300)).astype(np.float32)
m = V.SeqPAN(cfg, glove).to(dev); m.sync_timing = False; m.train()
opt = FlatAdamW(m, lr=1e-4)
batch = {k: v.to(dev) for k, v in Bn.synth(a, 1).items()}
def step():
    loss, out = V.train_engine_SeqPAN(m, batch, cfg, "train")
    opt.zero_grad(); loss.backward(); opt.step()
    return loss
for _ in range(3): step()
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/n:.2f} ms/step, total {1e3*(t2-t0)/n:.2f} ms/step")
# forward only / backward only enqueue split
t0 = time.perf_counter()
for _ in range(n):
    loss, out = V.train_engine_SeqPAN(m, batch, cfg, "train")
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"fwd-only enqueue {1e3*(t1-t0)/n:.2f} ms, total {1e3*(t2-t0)/n:.2f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(3): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
