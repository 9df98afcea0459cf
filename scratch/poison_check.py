"""Find reads of uninitialised memory: every torch.empty / empty_like the host layer makes is NaN-filled; the first
autograd Function (forward or backward) whose outputs carry a NaN inside their VALID region read padding it must not."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd import ops
from vmrframe_amd.optim import FlatAdamW
from tests.helpers import load_golden
from tests.test_gpu_trainer import build
dev = torch.device("cuda")
_empty, _empty_like = torch.empty, torch.empty_like
POISON = [True]
def p_empty(*a, **k):
    t = _empty(*a, **k)
    if POISON[0] and t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
    return t
def p_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if POISON[0] and t.is_floating_point() and t.is_cuda: t.fill_(float("nan"))
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
            seen.append(("fwd", cls.__name__, [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)][:3]))
        return out
    def b(ctx, *a, **k):
        out = bwd(ctx, *a, **k)
        outs = out if isinstance(out, tuple) else (out,)
        if any(has_nan(o) for o in outs) and not any(has_nan(x) for x in a):
            seen.append(("bwd", cls.__name__, [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)][:3]))
        return out
    cls.forward, cls.backward = staticmethod(f), staticmethod(b)
for name in dir(ops):
    c = getattr(ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
for gname, dtype in (("g_small", "bf16"), ("g_small", "fp32"), ("g_cfg2_small_B", "bf16"), ("g_masks", "fp32")):
    z, cfg, batch, g, weights = load_golden(gname)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    m = build(cfg, weights, dtype, dev, g, droprate=0.2, train=True)
    opt = FlatAdamW(m, lr=0.0, max_norm=1.0)
    for it in range(2):
        seen.clear()
        loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
        opt.zero_grad(); loss.backward()
        torch.cuda.synchronize()
        gn = float(sum(p.grad.double().pow(2).sum() for p in m.parameters() if p.grad is not None))
        print(gname, dtype, "pass", it, "loss", float(loss), "gradsq", gn, "first offenders:", seen[:6], flush=True)
        opt.step()
