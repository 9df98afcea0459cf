"""Which torch (non-library) ops are left in one eager BAN train step, by device time: op, input shapes, source line."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench as Bn
from types import SimpleNamespace as NS
import vmrframe_amd as V
from vmrframe_amd.optim import FlatAdamW

dev = torch.device("cuda", 0)
C5 = Bn.CFG5
B, T, Vw, E, Lq = C5["B"], C5["N"], 4000, 300, 20
cfg = NS(device=dev, model=NS(vlen=T, topk=20, neighbor=3, negative=0, prop_num=80, sparse_sample=True, pooling_counts=C5["pooling"],
                              fuse_dim=C5["F"], vdim=1024, dim=C5["F"] // 2, lstm_layer=2, query_embed_dim=E, contrast_dim=C5["Cd"],
                              droprate=0.1, gcn=NS(num_blocks=2, k=80, hidden_size=C5["F"])),
         loss=NS(min_iou=C5["min_iou"], max_iou=C5["max_iou"], bce=2.0, refine=1.0, td=0.1, offset=1.0, contrast=0.1))
rng = np.random.default_rng(1234)
model = V.BAN(cfg, pre_train_emb=rng.standard_normal((Vw, E)).astype(np.float32), compute_dtype=torch.bfloat16, sync_timing=False).to(dev).train()
opt = FlatAdamW(model, lr=1e-4, weight_decay=0.01, max_norm=1.0)
gen = torch.Generator().manual_seed(1234)
vl = torch.randint(T // 2, T + 1, (B,), generator=gen); vl[0] = T
ql = torch.randint(5, Lq + 1, (B,), generator=gen); ql[0] = Lq
data = {"vfeats": torch.randn(B, T, 1024, generator=gen), "words_ids": torch.randint(1, Vw + 2, (B, Lq), generator=gen),
        "vlens": vl, "tlens": ql, "start_end_offset": torch.randn(B, T, T, 2, generator=gen),
        "iou2ds": torch.rand(B, T, T, generator=gen), "dist_idxs": torch.rand(B, 2, T, generator=gen),
        "map2d_contrasts": torch.rand(B, 2, T, T, generator=gen) > 0.5}
data = {k: v.to(dev) for k, v in data.items()}


def step():
    opt.zero_grad()
    loss, _ = V.train_engine_BAN(model, data, cfg, "train")
    opt.backward(loss)
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=4)
rows = []
for e in ka:
    t = getattr(e, "self_device_time_total", None)
    if t is None:
        t = getattr(e, "self_cuda_time_total", 0)
    if t > 0 and e.key.startswith("aten::"):
        st = [s for s in e.stack if "vmrframe_amd" in s or "bench" in s or "scratch" in s]
        rows.append((t, e.count, e.key, str(e.input_shapes)[:90], st[0].split("/")[-1][:60] if st else ""))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print(f"aten device time in one eager step: {tot / 1e3:.2f} ms over {sum(r[1] for r in rows)} ops")
for t, n, k, shp, st in rows[:45]:
    print(f"{t:8.0f} us  x{n:<4d} {k:34s} {shp:90s} {st}")
