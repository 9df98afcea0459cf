"""BAN (row N2, BASELINE.json configs[4]: "BAN.py path, T=128 -> 128x128 score map") at config/anet/BAN.yaml's model sizes:
one train step = forward (incl. the host sampler round trip) + the five losses + backward + torch.optim.AdamW, B = 64,
bf16 compute.  Eager (the sampler's device-to-host copy rules a whole-step hipGraph out for now).  Also prints the step's
share of the host sampler."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from types import SimpleNamespace
import numpy as np
import torch
from vmrframe_amd.ban import BAN, train_engine_BAN
import vmrframe_amd.ban as banmod

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, V, E, Lq = int(os.environ.get("BAN_B", 64)), 128, 4000, 300, 20
cfg = SimpleNamespace(device=dev,
                      model=SimpleNamespace(vlen=T, topk=20, neighbor=3, negative=0, prop_num=80, sparse_sample=True,
                                            pooling_counts=[31, 16, 16], fuse_dim=512, vdim=1024, dim=256, lstm_layer=2,
                                            query_embed_dim=E, contrast_dim=128, droprate=0.1,
                                            gcn=SimpleNamespace(num_blocks=2, k=80, hidden_size=512)),
                      loss=SimpleNamespace(min_iou=0.5, max_iou=1.0, bce=1.0, refine=1.0, td=0.1, offset=1.0, contrast=0.1))
model = BAN(cfg, pre_train_emb=np.random.randn(V, E).astype(np.float32), compute_dtype=torch.bfloat16, sync_timing=False).to(dev).train()
opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4)
rng = np.random.default_rng(0)
vl = torch.randint(T // 2, T + 1, (B,)); vl[0] = T
ql = torch.randint(5, Lq + 1, (B,)); ql[0] = Lq
data = {"vfeats": torch.randn(B, T, 1024), "words_ids": torch.randint(1, V + 2, (B, Lq)), "vlens": vl, "tlens": ql,
        "start_end_offset": torch.randn(B, T, T, 2), "iou2ds": torch.rand(B, T, T), "dist_idxs": torch.rand(B, 2, T),
        "map2d_contrasts": torch.rand(B, 2, T, T) > 0.5}
data = {k: v.to(dev) for k, v in data.items()}
t_samp = [0.0]
orig = banmod.sample_proposals
def timed(*a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = orig(*a, **k); t_samp[0] += time.perf_counter() - t0; return r
banmod.sample_proposals = timed


def step():
    opt.zero_grad(set_to_none=True)
    loss, out = train_engine_BAN(model, data, cfg, "train")
    loss.backward()
    opt.step()
    return loss


for _ in range(3): step()
torch.cuda.synchronize(); t_samp[0] = 0.0
n = 10
t0 = time.perf_counter()
for _ in range(n): loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
print(f"BAN train step (anet sizes, B={B}, bf16, eager): {dt * 1e3:.1f} ms = {B / dt:.0f} clips/s; host sampler {t_samp[0] / n * 1e3:.1f} ms of it; loss {float(loss):.4f}")
