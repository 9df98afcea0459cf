"""One GEMM shape, eager launches, for a rocprofv3 --pmc pass: python3 gemm_pmc.py M N K [epi]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import _lib as L
M, N, K = (int(x) for x in sys.argv[1:4])
dev = "cuda"; dt = torch.bfloat16
nset = 4
As = [torch.randn(M, K, device=dev).to(dt) for _ in range(nset)]
Bs = [torch.randn(N, K, device=dev).to(dt) for _ in range(nset)]
Cs = [torch.zeros(M, N, device=dev, dtype=dt) for _ in range(nset)]
lib = L.lib(); st = torch.cuda.current_stream().cuda_stream
descs = []
for i in range(nset):
    d = L.GemmDesc(); d.A, d.B, d.C = As[i].data_ptr(), Bs[i].data_ptr(), Cs[i].data_ptr()
    d.lda, d.ldb, d.ldc, d.ldr = K, K, N, N
    d.M, d.N, d.K, d.transA, d.transB, d.dtype = M, N, K, 0, 0, 1
    d.flags = 0; d.alpha = 1.0; d.Z1 = d.Z2 = 1; d.splitk = 1
    descs.append(d)
for i in range(12):
    lib.vmr_gemm(C.byref(descs[i % nset]), st)
torch.cuda.synchronize()
