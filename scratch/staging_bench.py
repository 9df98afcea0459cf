"""Throughput of the device input staging (vmrframe_amd.staging.FeatureArena, row N3) next to the reference's
host loop restated in oracle/staging_ref.py, on a synthetic cfg2-shaped dataset (V=500, max_vlen=128)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from vmrframe_amd import staging
from oracle import staging_ref as S
rng = np.random.default_rng(1)
T, V, NV, B = 128, 500, 1500, 64
feats = {f"v{i}": rng.standard_normal((int(n), V)).astype(np.float32) for i, n in enumerate(rng.integers(30, 900, size=NV))}
arena = staging.FeatureArena(feats, T, "truncation")
ids = list(feats)
batches = [[ids[j] for j in rng.integers(0, NV, size=B)] for _ in range(40)]
for b in batches[:5]: arena.stage(b)
torch.cuda.synchronize(); t0 = time.time()
for b in batches: out = arena.stage(b)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"device staging: {len(batches)*B/dt:.0f} clips/s ({dt/len(batches)*1e3:.2f} ms per batch of {B}, arena {arena.arena.numel()*4/1e9:.2f} GB in HBM)")
tf = {k: torch.from_numpy(v) for k, v in feats.items()}
t0 = time.time()
for b in batches[:6]:
    vf, vm, vl = S.stage_batch([tf[i] for i in b], T, "truncation"); vf = vf.cuda()
torch.cuda.synchronize(); dt = time.time() - t0
print(f"host loop (oracle restatement of the reference, 1 process) + H2D: {6*B/dt:.0f} clips/s ({dt/6*1e3:.1f} ms per batch)")
