"""Is a hipGraph replay ordered after earlier work on torch's current (null) stream?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import _lib as L
dev = torch.device("cuda")
n = 64 << 20
x = torch.zeros(n, device=dev); y = torch.zeros(n, device=dev); src = torch.zeros(n, device=dev)
xb = torch.zeros(n, device=dev, dtype=torch.bfloat16)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y.copy_(x * 2)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y.copy_(x * 2)
g2 = torch.cuda.CUDAGraph()
yb = torch.zeros(n, device=dev)
with torch.cuda.graph(g2):
    yb.copy_(xb.float() * 2)
bad = {"fill": 0, "copy": 0, "ctypes_cast": 0}
for i in range(1, 41):
    x.fill_(float(i)); g.replay()
    if float(y[-1].item()) != 2.0 * i or float(y[0].item()) != 2.0 * i: bad["fill"] += 1
    src.fill_(float(i) + 0.5); torch.cuda.synchronize()
    x.copy_(src); g.replay()
    if float(y[-1].item()) != 2.0 * i + 1 or float(y[n // 2].item()) != 2.0 * i + 1: bad["copy"] += 1
    src.fill_(float(i)); torch.cuda.synchronize()
    L.check(L.lib().vmr_cast(src.data_ptr(), L.F32, xb.data_ptr(), L.BF16, n // 8, 8, 8, 8, 0.0, 0, None, L.stream_ptr()), "cast")
    g2.replay()
    if float(yb[-1].item()) != 2.0 * i or float(yb[n // 2].item()) != 2.0 * i: bad["ctypes_cast"] += 1
print("ordering violations out of 40:", bad, "current stream ptr", L.stream_ptr())
