import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd import ops
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
def P(*a): print(*a, flush=True)
dev = torch.device("cuda")
a = Bn.CFG2
cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
torch.manual_seed(0)
glove = np.random.default_rng(0).standard_normal((a["num_words"]-2, 300)).astype(np.float32)
m = V.SeqPAN(cfg, glove).to(dev); m.sync_timing = False; m.train()
opt = FlatAdamW(m, lr=1e-4)
batch = {k: v.to(dev) for k, v in Bn.synth(a, 1).items()}
g = GraphedTrainStep(m, opt, V.train_engine_SeqPAN, cfg, None, warmup=3).capture(batch)
torch.cuda.synchronize(); P("captured")
for i in range(3): l = g()
torch.cuda.synchronize(); P("3 replays ok", float(l.item()))
t0 = time.perf_counter()
for i in range(20): l = g()
torch.cuda.synchronize(); P("20 replays ok", float(l.item()), f"{(time.perf_counter()-t0)/20*1e3:.2f} ms/step")
def eager_step():
    loss, out = V.train_engine_SeqPAN(m, batch, cfg, "train")
    opt.zero_grad(); loss.backward(); opt.step(); return loss
l = eager_step(); torch.cuda.synchronize(); P("eager after graph ok", float(l.item()))
t = Bn.GemmTimer(); ops.GEMM_HOOK = t
l = eager_step(); torch.cuda.synchronize(); ops.GEMM_HOOK = None; P("eager with hook ok", t.summary())
for i in range(5): l = g()
torch.cuda.synchronize(); P("replays after eager ok", float(l.item()))
