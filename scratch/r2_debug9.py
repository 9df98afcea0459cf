"""Where does a bad replay's forward deviate from an eager forward on the same weights?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd import ops
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
from tests.helpers import load_golden
from tests.test_gpu_trainer import build, arena_grads
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
sched = dict(warmup_steps=0.0, total_steps=10)
REC = [None]
def wrap(cls):
    fwd = cls.forward
    def f(ctx, *a, **k):
        out = fwd(ctx, *a, **k)
        if REC[0] is not None:
            o = out if isinstance(out, tuple) else (out,)
            REC[0].append((cls.__name__, [t for t in o if isinstance(t, torch.Tensor)],
                           [t for t in a if isinstance(t, torch.Tensor)][:4]))
        return out
    cls.forward = staticmethod(f)
for name in dir(ops):
    c = getattr(ops, name)
    if isinstance(c, type) and issubclass(c, torch.autograd.Function) and c is not torch.autograd.Function:
        wrap(c)
A = build(cfg, weights, "bf16", dev, g)
optA = FlatAdamW(A, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
step = GraphedTrainStep(A, optA, V.train_engine_SeqPAN, cfg, warmup=2)
_fb = step._fwd_bwd
cnt = [0]
rec_graph = []
def fb():
    cnt[0] += 1
    REC[0] = rec_graph if cnt[0] == 3 else None
    r = _fb()
    REC[0] = None
    return r
step._fwd_bwd = fb
step.capture(batch)
stepA = step
B = build(cfg, weights, "bf16", dev, g)
optB = FlatAdamW(B, lr=1e-3, weight_decay=0.01, max_norm=1.0, **sched)
dbatch = {k: v.to(dev) for k, v in batch.items()}
def eager():
    loss, _ = V.train_engine_SeqPAN(B, dbatch, cfg, "train")
    optB.zero_grad(); loss.backward(); optB.step()
    return float(loss.item())
eager(); eager()
named = dict(B.named_parameters())
for it in range(3):
    for dst, src in ((optA.arena.flat_p, optB.arena.flat_p), (optA.m, optB.m), (optA.v, optB.v), (optA.step_t, optB.step_t)):
        dst.copy_(src)
    optA.sync_mirrors()
    before = optB.arena.flat_p.clone()
    wsnap = optA.arena.flat_p.clone()
    la = float(stepA().item())
    torch.cuda.synchronize()
    snap = [(n, [t.detach().clone() for t in outs], [t.detach().clone() for t in ins]) for n, outs, ins in rec_graph]
    after = (optA.arena.flat_p.clone(), optA.m.clone(), optA.v.clone(), optA.step_t.clone())
    # eager forward of A on the weights the replay started from
    optA.arena.flat_p.copy_(wsnap); optA.sync_mirrors()
    rec_e = []
    REC[0] = rec_e
    with torch.no_grad():
        le = float(V.train_engine_SeqPAN(A, dbatch, cfg, "train")[0].item())
    REC[0] = None
    optA.arena.flat_p.copy_(after[0]); optA.m.copy_(after[1]); optA.v.copy_(after[2]); optA.step_t.copy_(after[3]); optA.sync_mirrors()
    lb = eager()
    torch.cuda.synchronize()
    print(it, "replay", round(la, 4), "A eager same weights", round(le, 4), "B eager", round(lb, 4), "ops", len(snap), len(rec_e), flush=True)
    shown = 0
    for i, ((n1, o1, i1), (n2, o2, i2)) in enumerate(zip(snap, rec_e)):
        assert n1 == n2
        d = [float((x.float() - y.float()).abs().max()) for x, y in zip(o1, o2)]
        di = [float((x.float() - y.float()).abs().max()) if x.shape == y.shape else -1 for x, y in zip(i1, i2)]
        if max(d) > 1e-2 and shown < 6:
            print("    op", i, n1, "out diff", [round(v, 4) for v in d], "in diff", [round(v, 4) for v in di], [tuple(x.shape) for x in o1]); shown += 1
    # the same tensor loop as the test
    gB = arena_grads(optB, B)
    gmax = max(float(v.abs().max()) for v in gB.values())
    for n in optB.names:
        if float(gB[n].abs().max()) < 1e-4 * gmax:
            continue
        o, k = optB.offsets[n], named[n].numel()
        dA = (optA.arena.flat_p[o:o + k] - before[o:o + k]).double()
        dB = (optB.arena.flat_p[o:o + k] - before[o:o + k]).double()
        float((dA - dB).norm() / dB.norm())
