"""BaseFast goes NaN around step 32 of the bench loop: find the first autograd Function whose outputs carry a NaN/Inf
while its inputs are finite."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd import ops
from vmrframe_amd.optim import FlatAdamW
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
a = Bn.CFG4
torch.manual_seed(1234)
cfg = Bn.make_cfg(a, "bf16"); cfg.device = dev
glove = np.random.default_rng(1234).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
model = V.BaseFast(cfg, glove).to(dev); model.sync_timing = False; model.base_seed = 1234
opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=2100)
batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
model.train()
CHECK = [False]; seen = []
def bad(x):
    return isinstance(x, torch.Tensor) and x.is_floating_point() and not bool(torch.isfinite(x.float()).all())
def stat(x):
    return (tuple(x.shape), str(x.dtype)[6:], float(x.float().abs().max())) if isinstance(x, torch.Tensor) else None
def wrap(cls):
    fwd, bwd = cls.forward, cls.backward
    def f(ctx, *a, **k):
        out = fwd(ctx, *a, **k)
        if CHECK[0]:
            outs = out if isinstance(out, tuple) else (out,)
            if any(bad(o) for o in outs) and not any(bad(x) for x in a):
                seen.append(("fwd", cls.__name__, [stat(x) for x in a if isinstance(x, torch.Tensor)][:6]))
        return out
    def b(ctx, *a, **k):
        out = bwd(ctx, *a, **k)
        if CHECK[0]:
            outs = out if isinstance(out, tuple) else (out,)
            if any(bad(o) for o in outs) and not any(bad(x) for x in a):
                seen.append(("bwd", cls.__name__, [stat(x) for x in a if isinstance(x, torch.Tensor)][:6],
                             [stat(x) for x in ctx.saved_tensors][:8]))
        return out
    cls.forward, cls.backward = staticmethod(f), staticmethod(b)
for name in dir(ops):
    c = getattr(ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
for it in range(60):
    CHECK[0] = it >= 24
    loss, out = V.train_engine_BaseFast(model, batch, cfg, "train")
    opt.zero_grad(); loss.backward()
    gbad = opt.arena is not None and not bool(torch.isfinite(opt.arena.flat_g).all())
    opt.step()
    torch.cuda.synchronize()
    lv = float(loss.item())
    if it % 4 == 0 or seen or gbad or not np.isfinite(lv):
        print(it, lv, "grad-arena-nonfinite" if gbad else "", flush=True)
    if seen or gbad or not np.isfinite(lv):
        for s in seen[:6]: print("   ", s)
        if gbad:
            g = opt.arena.flat_g
            for n in opt.names:
                o = opt.offsets[n]; p = dict(model.named_parameters())[n]
                if not bool(torch.isfinite(g[o:o + p.numel()]).all()): print("    nonfinite grad:", n)
        break
