"""Reference point (NOT used by the product): what the vendor BLAS behind torch.matmul reaches on the step's GEMM shapes,
graph-timed with rotating operands like scratch/gemm_bench.py."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gemm_bench import bench
dev = "cuda"
def blas(M, N, K, iters=50):
    dt = torch.bfloat16
    nset = 6
    As = [torch.randn(M, K, device=dev).to(dt) for _ in range(nset)]
    Bs = [torch.randn(N, K, device=dev).to(dt) for _ in range(nset)]
    Cs = [torch.empty(M, N, device=dev, dtype=dt) for _ in range(nset)]
    for i in range(6): torch.mm(As[i % nset], Bs[i % nset].t(), out=Cs[i % nset])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): torch.mm(As[i % nset], Bs[i % nset].t(), out=Cs[i % nset])
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    print(f"torch.mm (vendor BLAS) M{M} N{N} K{K}: {t*1e6:7.1f} us {2*M*N*K/t/1e12:7.1f} TF", flush=True)
for (M, N, K) in [(8192, 1024, 1024), (9472, 1024, 1024), (9472, 3072, 1024), (8192, 1024, 4096), (4096, 4096, 4096), (8192, 8192, 8192)]:
    blas(M, N, K)
    bench(M, N, K, 0, 0, tag="ours")
