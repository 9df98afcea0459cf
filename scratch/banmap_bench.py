"""Row N2 measurement: the BAN proposal-map stage at BASELINE configs[4] shapes (B=64, N=128, F=512, contrast 128,
pooling_counts [31,16,16], bf16): fwd + loss_bce + bwd per step, and the two map2d kernels against HBM.  GPU box only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import ops, ban_map

dev = torch.device("cuda", 0)
B, N, F, Cd = 64, 128, 512, 128
torch.manual_seed(0)
m = ban_map.ProposalMap2D(F, Cd, N, [31, 16, 16]).to(dev).train()
hb = torch.relu(torch.randn(B, N, F, device=dev)).to(torch.bfloat16).requires_grad_(True)
fuse = torch.tanh(torch.randn(B, N, F, device=dev)).to(torch.bfloat16).requires_grad_(True)
iou = torch.rand(B, N, N, device=dev)
lay = m.layout.to(dev)
opt = torch.optim.AdamW(m.parameters(), lr=1e-4)

def step(dense=True):
    out = m(hb, fuse, dense_outputs=dense)
    loss = ban_map.bce_map_loss(out["tmap_cells"], iou, lay, 0.5, 1.0)
    if dense:
        loss = loss + 0.0 * out["map2d_proj"].float().mean()
    opt.zero_grad(set_to_none=True)
    hb.grad = None; fuse.grad = None
    loss.backward()
    opt.step()
    return loss

def timeit(fn, n=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

for dense in (True, False):
    t = timeit(lambda: step(dense))
    C = lay.C
    flops_fwd = 2 * B * (2 * N * F * F + C * (F * F + F * F + F * Cd + Cd * Cd + F))
    print(f"stage step (dense outputs {dense}): {t*1e3:.2f} ms  {B/t:.0f} clips/s  ~{3*flops_fwd/t/1e12:.0f} TFLOP/s (3x fwd flops {flops_fwd/1e9:.0f} GF)", flush=True)

x = fuse.detach(); ps = torch.randn(B * N, F, device=dev).to(torch.bfloat16); pe = torch.randn_like(ps)
C = lay.C
tf = timeit(lambda: ops.map2d_pool(x, ps, pe, lay), 20)
byt = (3 * B * N * F + 2 * B * C * F) * 2
print(f"map2d_pool_fwd: {tf*1e6:.1f} us  {byt/tf/1e12:.2f} TB/s  (algorithmic {byt/1e6:.0f} MB)")
xr = x.clone().requires_grad_(True); psr = ps.clone().requires_grad_(True); per = pe.clone().requires_grad_(True)
M, R = ops.map2d_pool(xr, psr, per, lay)
gM, gR = torch.randn_like(M), torch.randn_like(R)
tb = timeit(lambda: torch.autograd.grad([M, R], [xr, psr, per], [gM, gR], retain_graph=True), 20)
bytb = (2 * B * C * F + 4 * B * N * F) * 2
print(f"map2d_pool_bwd (+dp): {tb*1e6:.1f} us  {bytb/tb/1e12:.2f} TB/s  (algorithmic {bytb/1e6:.0f} MB)")
if "--no-torch-prof" in sys.argv:
    sys.exit(0)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(True); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
