import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import ops, _lib as L
dev="cuda"; dt=torch.bfloat16
def timeit(name, fn, bytes_, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)*1e-3/iters
    print(f"{name:28s} {t*1e6:8.1f} us  {bytes_/t/1e12:6.2f} TB/s (algorithmic {bytes_/1e6:.1f} MB)", flush=True)
N, D = 9472, 1024
lib=L.lib(); st=lambda: torch.cuda.current_stream().cuda_stream
x=torch.randn(N,D,device=dev).to(dt); dy=torch.randn(N,D,device=dev).to(dt); y=torch.empty_like(x); dx=torch.empty_like(x)
g=torch.ones(D,device=dev); b=torch.zeros(D,device=dev); mean=torch.zeros(N,device=dev); rstd=torch.ones(N,device=dev)
dg=torch.zeros(D,device=dev); db=torch.zeros(D,device=dev); ws=torch.empty(4096*2*D,device=dev)
timeit("ln_fwd", lambda: lib.vmr_layernorm_fwd(x.data_ptr(),g.data_ptr(),b.data_ptr(),1e-6,None,0,y.data_ptr(),mean.data_ptr(),rstd.data_ptr(),N,D,1,0.0,0,None,st()), 2*N*D*2)
timeit("ln_fwd+dropout", lambda: lib.vmr_layernorm_fwd(x.data_ptr(),g.data_ptr(),b.data_ptr(),1e-6,None,0,y.data_ptr(),mean.data_ptr(),rstd.data_ptr(),N,D,1,0.2,5,None,st()), 2*N*D*2)
timeit("ln_bwd", lambda: lib.vmr_layernorm_bwd(dy.data_ptr(),x.data_ptr(),g.data_ptr(),mean.data_ptr(),rstd.data_ptr(),None,dx.data_ptr(),dg.data_ptr(),db.data_ptr(),None,ws.data_ptr(),0,N,D,1,0.0,0,None,st()), 3*N*D*2)
timeit("ln_bwd+dres+dropout", lambda: lib.vmr_layernorm_bwd(dy.data_ptr(),x.data_ptr(),g.data_ptr(),mean.data_ptr(),rstd.data_ptr(),dy.data_ptr(),dx.data_ptr(),dg.data_ptr(),db.data_ptr(),None,ws.data_ptr(),0,N,D,1,0.2,5,None,st()), 4*N*D*2)
w=torch.randn(D,7,device=dev); dw=torch.zeros(D,7,device=dev)
def lnconv():
    lib.vmr_ln_dwconv_fwd(x.data_ptr(),g.data_ptr(),b.data_ptr(),1e-6,w.data_ptr(),y.data_ptr(),mean.data_ptr(),rstd.data_ptr(),64,128,D,1,st())
    lib.vmr_ln_dwconv_fwd(x[8192:].data_ptr(),g.data_ptr(),b.data_ptr(),1e-6,w.data_ptr(),y[8192:].data_ptr(),mean[8192:].data_ptr(),rstd[8192:].data_ptr(),64,20,D,1,st())
timeit("ln_dwconv_fwd (v+t)", lnconv, 2*N*D*2)
def dwb():
    lib.vmr_dwconv_bwd(dy.data_ptr(),x.data_ptr(),g.data_ptr(),b.data_ptr(),mean.data_ptr(),rstd.data_ptr(),w.data_ptr(),dx.data_ptr(),dw.data_ptr(),ws.data_ptr(),64,128,D,1,st())
    lib.vmr_dwconv_bwd(dy[8192:].data_ptr(),x[8192:].data_ptr(),g.data_ptr(),b.data_ptr(),mean[8192:].data_ptr(),rstd[8192:].data_ptr(),w.data_ptr(),dx[8192:].data_ptr(),dw.data_ptr(),ws.data_ptr(),64,20,D,1,st())
timeit("dwconv_bwd (v+t)", dwb, 3*N*D*2)
h=torch.randn(N,D,device=dev).to(dt)
timeit("relu_bwd_bias mode0", lambda: lib.vmr_relu_bwd_bias(0,dy.data_ptr(),None,None,db.data_ptr(),N,D,D,1.0,1,0.0,0,None,None,1.0,st()), N*D*2)
timeit("relu_bwd_bias mode1", lambda: lib.vmr_relu_bwd_bias(1,dy.data_ptr(),h.data_ptr(),dx.data_ptr(),db.data_ptr(),N,D,D,1.25,1,0.0,0,None,None,1.0,st()), 3*N*D*2)
timeit("relu_bwd_bias mode2", lambda: lib.vmr_relu_bwd_bias(2,dy.data_ptr(),None,dx.data_ptr(),db.data_ptr(),N,D,D,1.25,1,0.2,5,None,None,1.0,st()), 2*N*D*2)
B,H,T=64,4,128
S=torch.randn(B,H,T,T,device=dev); P=torch.empty(B,H,T,T,device=dev,dtype=dt); Pk=torch.empty_like(P); vm=torch.ones(B,T,device=dev)
timeit("softmax_fwd self (no drop)", lambda: lib.vmr_softmax_fwd(S.data_ptr(),P.data_ptr(),None,vm.data_ptr(),vm.data_ptr(),0,B*H,H,T,T,T,T,0,0.0625,1,0.0,0,None,st()), B*H*T*T*6)
timeit("softmax_fwd self (drop)", lambda: lib.vmr_softmax_fwd(S.data_ptr(),P.data_ptr(),Pk.data_ptr(),vm.data_ptr(),vm.data_ptr(),0,B*H,H,T,T,T,T,0,0.0625,1,0.2,7,None,st()), B*H*T*T*8)
dS=torch.empty_like(P)
timeit("softmax_bwd self (drop)", lambda: lib.vmr_softmax_bwd(S.data_ptr(),Pk.data_ptr(),dS.data_ptr(),B*H,T,T,T,T,0.0625,1,0.2,7,None,st()), B*H*T*T*8)
pm=torch.zeros(65_000_000,device=dev); gg=torch.randn_like(pm); m1=torch.zeros_like(pm); v1=torch.zeros_like(pm); dec=torch.ones(pm.numel(),device=dev,dtype=torch.uint8); gs=torch.zeros(1,device=dev)
timeit("sumsq 65M", lambda: lib.vmr_sumsq(gg.data_ptr(),gs.data_ptr(),pm.numel(),st()), pm.numel()*4, iters=5)
timeit("adamw 65M", lambda: lib.vmr_adamw(pm.data_ptr(),gg.data_ptr(),m1.data_ptr(),v1.data_ptr(),dec.data_ptr(),None,0,gs.data_ptr(),1.0,1e-4,0.9,0.999,1e-8,0.01,1,None,0.0,0.0,None,pm.numel(),st()), pm.numel()*(7*4+1), iters=5)
