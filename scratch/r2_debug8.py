"""Which device pointers baked into the captured step graph point into general-pool memory that is FREE after capture?"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vmrframe_amd as V
from vmrframe_amd import ops, _lib as L
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep
from tests.helpers import load_golden
from tests.test_gpu_trainer import build
dev = torch.device("cuda")
z, cfg, batch, g, weights = load_golden("g_small")
A = build(cfg, weights, "bf16", dev, g)
optA = FlatAdamW(A, lr=1e-3, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=10)
log = []
REC = [False]
h = L.lib()
def mk(name, fn):
    def w(*a):
        if REC[0]:
            for i, x in enumerate(a):
                if isinstance(x, int) and x > (1 << 32):
                    log.append((name, i, x))
        return fn(*a)
    return w
for name in L.SIGNATURES:
    setattr(h, name, mk(name, getattr(h, name)))
_gemm = ops.gemm
def gemm_logged(A_, B_, C_, *a, **k):
    if REC[0]:
        for nm, t in (("A", A_), ("B", B_), ("C", C_), ("bias", k.get("bias")), ("residual", k.get("residual")), ("aux", k.get("aux")),
                      ("rowscale", k.get("rowscale")), ("bias2", k.get("bias2")), ("a_colsum", k.get("a_colsum"))):
            if t is not None:
                log.append(("gemm." + nm, 0, t.data_ptr()))
    return _gemm(A_, B_, C_, *a, **k)
ops.gemm = gemm_logged
step = GraphedTrainStep(A, optA, V.train_engine_SeqPAN, cfg, warmup=2)
_fb = step._fwd_bwd
cnt = [0]
def fb():
    cnt[0] += 1
    REC[0] = cnt[0] == 3        # the third call is the captured one
    r = _fb()
    return r
step._fwd_bwd = fb
_st = optA.step
def st():
    r = _st(); 
    if cnt[0] == 3: REC[0] = False
    return r
optA.step = st
step.capture(batch)
torch.cuda.synchronize()
snap = torch.cuda.memory_snapshot()
blocks = []
for s in snap:
    addr = s["address"]
    for b in s["blocks"]:
        blocks.append((addr, addr + b["size"], b["state"], tuple(s.get("segment_pool_id", (0, 0))), s.get("stream")))
        addr += b["size"]
def find(p):
    for a, b, stt, pid, strm in blocks:
        if a <= p < b:
            return stt, pid, strm
    return ("unknown", None, None)
from collections import Counter
c = Counter()
bad = []
for name, i, p in log:
    stt, pid, strm = find(p)
    c[(stt, pid == (0, 0))] += 1
    if stt != "active_allocated":
        bad.append((name, i, hex(p), stt, pid))
print("pointers logged:", len(log), "by (state, general_pool):", dict(c))
seen = set()
for b in bad:
    if (b[0], b[1]) not in seen:
        seen.add((b[0], b[1])); print("  NOT LIVE:", b)
