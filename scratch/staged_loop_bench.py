"""The captured cfg2 train step fed a FRESH batch every step by staging.BatchStager (device-resident feature arena,
text / label collate on the host into a pinned double buffer, one async copy per batch on a copy stream) against the
same step on a resident batch.  usage: staged_loop_bench.py [steps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vmrframe_amd as V
from vmrframe_amd import staging, synth
from vmrframe_amd.optim import FlatAdamW
from vmrframe_amd.trainer import GraphedTrainStep


def run(steps=200, B=64, T=128, Lq=20, D=1024, Vd=500, nvid=2000, seed=0, verbose=True):
    dev = torch.device("cuda")
    rng = np.random.default_rng(seed)
    cfg = synth.make_cfg(dim=D, vlen=T, vdim=Vd, num_words=4002, num_chars=60)
    cfg.model.compute_dtype = "bf16"; cfg.model.droprate = 0.2; cfg.device = dev
    glove = rng.standard_normal((4000, 300)).astype(np.float32)
    model = V.SeqPAN(cfg, glove).to(dev)
    model.sync_timing = False
    model.train()
    # a synthetic dataset in the reference's formats: <vid>.npy-like feature arrays of ragged length, word / char id lists
    vlens = rng.integers(T // 2, 3 * T, size=nvid)
    feats = {f"v{i}": rng.standard_normal((int(n), Vd)).astype(np.float32) for i, n in enumerate(vlens)}
    arena = staging.FeatureArena(feats, T, "truncation")
    wl = rng.integers(3, Lq + 1, size=nvid)
    wids = [rng.integers(2, 4002, size=int(n)).tolist() for n in wl]
    cids = [[rng.integers(1, 60, size=int(rng.integers(1, 9))).tolist() for _ in range(int(n))] for n in wl]
    out_len = np.minimum(vlens, T)
    s = (rng.random(nvid) * out_len).astype(np.int64); e = s + (rng.random(nvid) * (out_len - s)).astype(np.int64)
    st = staging.BatchStager(arena, staging.TextArena(wids, cids), list(feats), np.stack([s, e], 1), static_L=Lq, static_C=8)
    opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0)
    st.prefetch(rng.integers(0, nvid, size=B)); first = st.next()
    step = GraphedTrainStep(model, opt, V.train_engine_SeqPAN, cfg, warmup=3).capture(first)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    resident = B * steps / (time.perf_counter() - t0)
    st.prefetch(rng.integers(0, nvid, size=B))
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(steps):
        batch = st.next()
        step(batch)                                   # load_batch: device-to-device copies into the graph's static buffers
        h0 = time.perf_counter()
        st.prefetch(rng.integers(0, nvid, size=B))    # the next batch's host work + copy, under this step's replay
        host += time.perf_counter() - h0
    st.next()
    torch.cuda.synchronize()
    staged = B * steps / (time.perf_counter() - t0)
    loss = float(step.loss.item())
    if verbose:
        print(f"resident batch: {resident:8.1f} clips/s   fresh staged batch every step: {staged:8.1f} clips/s "
              f"({staged / resident * 100:.1f} %); host collate + enqueue {host / steps * 1e3:.3f} ms per batch; loss {loss:.3f}")
    return resident, staged, loss


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
