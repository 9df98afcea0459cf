"""BaseFast: NaN-fill every torch.empty the host layer makes; the first Function whose outputs carry a NaN while its
inputs do not reads memory it never wrote."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
import vmrframe_amd as V
from vmrframe_amd import ops
from vmrframe_amd.optim import FlatAdamW
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
_empty, _empty_like = torch.empty, torch.empty_like
def p_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
def p_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
torch.empty, torch.empty_like = p_empty, p_empty_like
seen = []
def has_nan(x):
    return isinstance(x, torch.Tensor) and x.is_floating_point() and bool(torch.isnan(x.float()).any())
def wrap(cls):
    fwd, bwd = cls.forward, cls.backward
    def f(ctx, *a, **k):
        out = fwd(ctx, *a, **k)
        outs = out if isinstance(out, tuple) else (out,)
        if any(has_nan(o) for o in outs) and not any(has_nan(x) for x in a):
            seen.append(("fwd", cls.__name__, [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)][:4], [tuple(o.shape) for o in outs if has_nan(o)]))
        return out
    def b(ctx, *a, **k):
        out = bwd(ctx, *a, **k)
        outs = out if isinstance(out, tuple) else (out,)
        if any(has_nan(o) for o in outs) and not any(has_nan(x) for x in a):
            seen.append(("bwd", cls.__name__, [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)][:4], [tuple(o.shape) for o in outs if has_nan(o)]))
        return out
    cls.forward, cls.backward = staticmethod(f), staticmethod(b)
for name in dir(ops):
    c = getattr(ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
for wl, a0, dtype in (("basefast", Bn.CFG4, "bf16"), ("basefast", Bn.CFG2, "fp32"), ("seqpan", Bn.CFG2, "bf16")):
    a = dict(a0); a["B"] = 8
    torch.manual_seed(1234)
    cfg = Bn.make_cfg(a, dtype); cfg.device = dev
    glove = np.random.default_rng(1234).standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
    Model, engine = (V.BaseFast, V.train_engine_BaseFast) if wl == "basefast" else (V.SeqPAN, V.train_engine_SeqPAN)
    model = Model(cfg, glove).to(dev); model.sync_timing = False; model.base_seed = 1234
    opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0)
    batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
    model.train()
    for it in range(4):
        seen.clear()
        loss, out = engine(model, batch, cfg, "train")
        opt.zero_grad(); loss.backward()
        torch.cuda.synchronize()
        gbad = opt.arena is not None and bool(torch.isnan(opt.arena.flat_g).any())
        print(wl, dtype, "pass", it, "loss", float(loss.item()), "arena grad NaN" if gbad else "", "offenders:", seen[:5], flush=True)
        opt.step()
        if opt.arena is not None and bool(torch.isnan(opt.arena.flat_p).any()):
            print("   params NaN after step"); break
