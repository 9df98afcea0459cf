import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import _lib as L
dev="cuda"; dt=torch.bfloat16
N, D = 9472, 1024
lib=L.lib(); st=lambda: torch.cuda.current_stream().cuda_stream
x=torch.randn(N,D,device=dev).to(dt); dy=torch.randn(N,D,device=dev).to(dt); y=torch.empty_like(x); dx=torch.empty_like(x)
g=torch.ones(D,device=dev); b=torch.zeros(D,device=dev); mean=torch.zeros(N,device=dev); rstd=torch.ones(N,device=dev)
dg=torch.zeros(D,device=dev); db=torch.zeros(D,device=dev); ws=torch.empty(8192*2*D,device=dev)
w=torch.randn(D,7,device=dev); dw=torch.zeros(D,7,device=dev)
B,H,T=64,4,128
S=torch.randn(B,H,T,T,device=dev); P=torch.empty(B,H,T,T,device=dev,dtype=dt); vm=torch.ones(B,T,device=dev)
for _ in range(3):
    lib.vmr_layernorm_fwd(x.data_ptr(),g.data_ptr(),b.data_ptr(),1e-6,None,0,y.data_ptr(),mean.data_ptr(),rstd.data_ptr(),N,D,1,0.0,0,None,st())
    lib.vmr_layernorm_bwd(dy.data_ptr(),x.data_ptr(),g.data_ptr(),mean.data_ptr(),rstd.data_ptr(),None,dx.data_ptr(),dg.data_ptr(),db.data_ptr(),None,ws.data_ptr(),0,N,D,1,0.0,0,None,st())
    lib.vmr_ln_dwconv_fwd(x.data_ptr(),g.data_ptr(),b.data_ptr(),1e-6,w.data_ptr(),y.data_ptr(),mean.data_ptr(),rstd.data_ptr(),64,128,D,1,st())
    lib.vmr_dwconv_bwd(dy.data_ptr(),x.data_ptr(),g.data_ptr(),b.data_ptr(),mean.data_ptr(),rstd.data_ptr(),w.data_ptr(),dx.data_ptr(),dw.data_ptr(),ws.data_ptr(),64,128,D,1,st())
    lib.vmr_softmax_fwd(S.data_ptr(),P.data_ptr(),None,vm.data_ptr(),vm.data_ptr(),0,B*H,H,T,T,T,T,0,0.0625,1,0.0,0,None,st())
    lib.vmr_relu_bwd_bias(1,dy.data_ptr(),x.data_ptr(),dx.data_ptr(),db.data_ptr(),N,D,D,1.25,1,0.0,0,None,st())
torch.cuda.synchronize()
