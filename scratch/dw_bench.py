import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import ops, _lib as L
dev="cuda"; dt=torch.bfloat16
def bench(M,N,K,sk,iters=40):
    nset=6
    As=[torch.randn(K,M,device=dev).to(dt) for _ in range(nset)]
    Bs=[torch.randn(K,N,device=dev).to(dt) for _ in range(nset)]
    ws=[torch.empty(sk,M,N,device=dev) for _ in range(nset)]
    def run(i):
        ops.gemm(As[i%nset],Bs[i%nset],ws[i%nset],M,N,K,1,1,M,N,N,dtype=L.BF16,flags=L.EPI_SLAB,splitk=sk)
    for i in range(6): run(i)
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): run(i)
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)*1e-3/iters
    print(f"DMA={os.environ.get('VMR_GEMM_DMA','2')} dW M{M} N{N} K{K} sk{sk}: {t*1e6:6.1f} us {2*M*N*K/t/1e12:6.1f} TF",flush=True)
for sk in (4,8,16):
    bench(1024,1024,9472,sk)
bench(1024,1024,8192,8)
bench(3072,1024,9472,2); bench(3072,1024,9472,4)

def bench_nt(M,N,K,sk,iters=40):
    nset=6
    As=[torch.randn(M,K,device=dev).to(dt) for _ in range(nset)]
    Bs=[torch.randn(N,K,device=dev).to(dt) for _ in range(nset)]
    ws=[torch.empty(sk,M,N,device=dev) for _ in range(nset)]
    def run(i):
        ops.gemm(As[i%nset],Bs[i%nset],ws[i%nset],M,N,K,0,0,K,K,N,dtype=L.BF16,flags=L.EPI_SLAB,splitk=sk)
    for i in range(6): run(i)
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): run(i)
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)*1e-3/iters
    print(f"NT-slab M{M} N{N} K{K} sk{sk}: {t*1e6:6.1f} us {2*M*N*K/t/1e12:6.1f} TF",flush=True)
bench_nt(1024,1024,9472,8)
bench_nt(1024,1024,8192,8)
