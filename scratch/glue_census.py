"""Census of the torch (non-library) ops in one SeqPAN train step: op name, shapes, dtype, count.  GPU box only."""
import collections, os, sys
import numpy as np, torch
from torch.utils._python_dispatch import TorchDispatchMode
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn

class Census(TorchDispatchMode):
    def __init__(self):
        super().__init__(); self.c = collections.Counter(); self.phase = "fwd"
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if any(k in name for k in ("view", "reshape", "as_strided", "detach", "alias", "t.default", "transpose", "slice", "select", "unsqueeze", "squeeze", "expand", "permute", "empty", "_unsafe_view", "is_", "sym_", "size", "stride", "unbind", "split", "narrow", "lift_fresh")):
            return out
        shp = tuple(tuple(a.shape) for a in args if isinstance(a, torch.Tensor))
        dt = tuple(str(a.dtype).replace("torch.", "") for a in args if isinstance(a, torch.Tensor))
        self.c[(self.phase, name, shp, dt)] += 1
        return out

def main():
    import vmrframe_amd as V
    from vmrframe_amd import dp
    from vmrframe_amd.optim import FlatAdamW
    dev = torch.device("cuda", 0); torch.cuda.set_device(0)
    a = Bn.CFG2; cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
    glove = np.random.default_rng(1234).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
    torch.manual_seed(1234)
    model = V.SeqPAN(cfg, glove).to(dev); model.sync_timing = False
    opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=100)
    batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
    model.train()
    def step(cs=None):
        if cs: cs.phase = "fwd"
        loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
        opt.zero_grad()
        if cs: cs.phase = "bwd"
        loss.backward()
        if cs: cs.phase = "opt"
        opt.step()
    for _ in range(2): step()
    cs = Census()
    with cs: step(cs)
    torch.cuda.synchronize()
    for (ph, name, shp, dt), n in sorted(cs.c.items(), key=lambda kv: (kv[0][0], -kv[1])):
        print(f"{ph} {n:3d} {name:40s} {shp} {dt}")
main()
