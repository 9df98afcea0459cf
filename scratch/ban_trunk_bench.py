"""BAN forward + backward up to the 2-D map at config/anet/BAN.yaml's sizes (vdim 1024, dim 256, lstm_layer 2, fuse_dim 512,
vlen 128, B = 64, 20-word queries, bf16, train mode): trunk (ban_trunk.BANTrunk) and trunk + map stage (ban_map.ProposalMap2D),
eager and hipGraph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from vmrframe_amd.ban_trunk import BANTrunk
from vmrframe_amd.ban_map import ProposalMap2D

dev = torch.device("cuda:0")
torch.manual_seed(0)
B, T, vdim, dim, F, Lq, V, E = 64, 128, 1024, 256, 512, 20, 4000, 300
trunk = BANTrunk(V + 2, vdim, dim, 2, E, F, T, np.random.randn(V, E).astype(np.float32), compute_dtype=torch.bfloat16).to(dev).train()
pm = ProposalMap2D(F, 128, T, [31, 16, 16], compute_dtype=torch.bfloat16).to(dev).train()
x = torch.randn(B, T, vdim, device=dev).bfloat16().requires_grad_(True)
vl = torch.randint(T // 2, T + 1, (B,), device=dev); vl[0] = T
ql = torch.randint(5, Lq + 1, (B,), device=dev); ql[0] = Lq
tok = torch.randint(1, V + 2, (B, Lq), device=dev)


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def step_trunk():
    o = trunk(x, tok, vl, ql, max_qlen=Lq)
    (o["fuse_feature"].float().square().mean() + o["hidden_b"].float().mean() + o["td"].mean() + o["sentence_feature"].float().sum()).backward()


def step_all():
    o = trunk(x, tok, vl, ql, max_qlen=Lq)
    r = pm(o["hidden_b"], o["fuse_feature"])
    (r["tmap"].float().square().mean() + r["map2d_proj"].float().square().mean() + o["td"].mean() + o["sentence_feature"].float().sum()).backward()


for name, fn in (("trunk", step_trunk), ("trunk + 2-D map stage", step_all)):
    t_e = timeit(fn, 3)
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(st); torch.cuda.synchronize()
    try:
        with torch.cuda.graph(g, stream=st):
            fn()
        t_g = timeit(g.replay, 10)
    except Exception as e:
        t_g = float("nan"); print("  graph capture failed:", str(e)[:200])
    print(f"{name}: fwd+bwd eager {t_e:.2f} ms, hipGraph replay {t_g:.2f} ms = {B / t_g * 1e3:.0f} clips/s", flush=True)
