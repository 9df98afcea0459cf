import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import ops, _lib as L
dev = "cuda"; dt = torch.bfloat16
N, D = 9472, 1024
lib = L.lib(); st = lambda: torch.cuda.current_stream().cuda_stream
nset = 8
sets = []
for _ in range(nset):
    sets.append(dict(x=torch.randn(N, D, device=dev).to(dt), dy=torch.randn(N, D, device=dev).to(dt),
                     dres=torch.randn(N, D, device=dev).to(dt), dx=torch.empty(N, D, device=dev, dtype=dt)))
g = torch.ones(D, device=dev); mean = torch.zeros(N, device=dev); rstd = torch.ones(N, device=dev)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev); ws = torch.empty(4096 * 2 * D, device=dev)
def run(i, dres=True, p=0.2):
    s = sets[i % nset]
    lib.vmr_layernorm_bwd(s["dy"].data_ptr(), s["x"].data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                          s["dres"].data_ptr() if dres else None, s["dx"].data_ptr(), dg.data_ptr(), db.data_ptr(), None,
                          ws.data_ptr(), 0, N, D, 1, p, 5, None, st())
for name, kw in (("ln_bwd+dres+drop", dict(dres=True, p=0.2)), ("ln_bwd plain", dict(dres=False, p=0.0))):
    for i in range(8): run(i, **kw)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(24): run(i, **kw)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    print(f"VMR_LNB_GRID={os.environ.get('VMR_LNB_GRID','dflt')} {name}: {e0.elapsed_time(e1)*1e3/24:.1f} us (ln_bwd + colreduce), cold operands")
