"""GEMM census of one SeqPAN train step: every vmr_gemm launch timed with HIP events (eager),
grouped by (M, N, K, ta, tb, Z); prints count/step, avg us, TFLOP/s, share.  GPU box only."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn  # noqa: E402


def main():
    import vmrframe_amd as V
    from vmrframe_amd import dp, ops
    from vmrframe_amd.optim import FlatAdamW
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    a = Bn.CFG2
    cfg = Bn.make_cfg(a, "bf16")
    cfg.device = dev
    rng = np.random.default_rng(1234)
    glove = rng.standard_normal((a["num_words"] - 2, 300)).astype(np.float32)
    torch.manual_seed(1234)
    model = V.SeqPAN(cfg, glove).to(dev)
    model.sync_timing = False
    opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0, warmup_steps=0.0, total_steps=100)
    batch = {k: v.to(dev) for k, v in Bn.synth(a, 1234).items()}
    model.train()

    def step():
        loss, out = V.train_engine_SeqPAN(model, batch, cfg, "train")
        opt.zero_grad()
        loss.backward()
        opt.step()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    rec = []

    def hook(launch, M, N, K, ta, tb, Z, dtype):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); launch(); e.record()
        rec.append(((M, N, K, ta, tb, Z, dtype), s, e))
    ops.GEMM_HOOK = hook
    nsteps = 3
    for _ in range(nsteps):
        step()
    torch.cuda.synchronize()
    ops.GEMM_HOOK = None
    agg = collections.OrderedDict()
    for key, s, e in rec:
        t = s.elapsed_time(e) * 1e3
        c = agg.setdefault(key, [0, 0.0])
        c[0] += 1; c[1] += t
    tot = sum(c[1] for c in agg.values())
    print(f"total GEMM event time per step: {tot / nsteps:.1f} us over {len(rec) / nsteps:.0f} launches")
    print(f"{'M':>6} {'N':>6} {'K':>6} ta tb {'Z':>5} dt {'n/step':>6} {'avg us':>8} {'TF/s':>7} {'us/step':>8} {'%':>5}")
    for key, c in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        M, N, K, ta, tb, Z, dtype = key
        avg = c[1] / c[0]
        tf = 2.0 * M * N * K * Z / (avg * 1e-6) / 1e12
        print(f"{M:6d} {N:6d} {K:6d} {ta:2d} {tb:2d} {Z:5d} {dtype:2d} {c[0] / nsteps:6.1f} {avg:8.1f} {tf:7.1f} "
              f"{c[1] / nsteps:8.1f} {100 * c[1] / tot:5.1f}")


if __name__ == "__main__":
    main()
