"""Find kernels whose result depends on what the PREVIOUS kernel left in LDS: every autograd Function of the host
layer runs twice on the same inputs, once after the LDS of all CUs was filled with NaN patterns and once after
zeros; outputs must be bit-equal and NaN-free."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd import ops, _lib as L
from vmrframe_amd.optim import FlatAdamW
from tests.helpers import load_golden
from tests.test_gpu_trainer import build
dev = torch.device("cuda")
scratch = torch.zeros(1, device=dev, dtype=torch.int32)
def poison(pat):
    L.check(L.lib().vmr_debug_poison_lds(pat, scratch.data_ptr(), L.stream_ptr()), "poison")
seen = []
def tens(o):
    o = o if isinstance(o, tuple) else (o,)
    return [t for t in o if isinstance(t, torch.Tensor)]
def same(a, b):
    if a.dtype in (torch.float32, torch.bfloat16, torch.float64):
        return bool(torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))) and not bool(torch.isnan(a.float()).any())
    return bool(torch.equal(a, b))
ACTIVE = [True]
def wrap(cls):
    fwd = cls.forward
    def f(ctx, *a, **k):
        if not ACTIVE[0]:
            return fwd(ctx, *a, **k)
        poison(0x7FC07FC0)
        o1 = [t.clone() for t in tens(fwd(ctx, *a, **k))]
        poison(0x00000000)
        out = fwd(ctx, *a, **k)
        o2 = tens(out)
        bad = [i for i, (x, y) in enumerate(zip(o1, o2)) if not same(x, y)]
        if bad:
            seen.append((cls.__name__, bad, [tuple(x.shape) for x in a if isinstance(x, torch.Tensor)][:3],
                         [float((o1[i].float() - o2[i].float()).abs().nan_to_num(nan=1e9).max()) for i in bad]))
        return out
    cls.forward = staticmethod(f)
for name in dir(ops):
    c = getattr(ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
for gname, dtype in (("g_small", "bf16"), ("g_small", "fp32"), ("g_cfg2_small_B", "bf16"), ("g_tiny", "fp32")):
    z, cfg, batch, g, weights = load_golden(gname)
    dbatch = {k: v.to(dev) for k, v in batch.items()}
    m = build(cfg, weights, dtype, dev, g, droprate=0.2, train=True)
    opt = FlatAdamW(m, lr=0.0, max_norm=1.0)
    for it in range(2):
        seen.clear()
        ACTIVE[0] = True
        loss, out = V.train_engine_SeqPAN(m, dbatch, cfg, "train")
        ACTIVE[0] = False
        opt.zero_grad(); loss.backward(); opt.step()
        torch.cuda.synchronize()
        print(gname, dtype, "pass", it, "loss", float(loss.detach()), "LDS-dependent forward ops:", len(seen), flush=True)
        for s_ in seen[:12]:
            print("    ", s_, flush=True)
