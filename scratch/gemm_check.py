import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import _lib as L
h = C.CDLL(L.LIB_PATH)
h.vmr_gemm.argtypes = [C.POINTER(L.GemmDesc), C.c_void_p]; h.vmr_gemm.restype = C.c_int
h.vmr_last_error.restype = C.c_char_p
dev = "cuda"

def gemm(A, B, ta, tb, M, N, K, flags=0, bias=None, res=None, aux=None, out_f32=False, alpha=1.0, splitk=1, Cacc=None, drop=(0.0, 0)):
    d = L.GemmDesc()
    d.A, d.B = A.data_ptr(), B.data_ptr()
    dt = L.dtype_code(A)
    if Cacc is not None:
        Cm = Cacc
    else:
        Cm = torch.empty(M, N, device=dev, dtype=torch.float32 if (out_f32 or dt == 0) else torch.bfloat16)
    d.C = Cm.data_ptr()
    d.lda, d.ldb, d.ldc, d.ldr = A.stride(0), B.stride(0), Cm.stride(0), N
    d.M, d.N, d.K, d.transA, d.transB, d.dtype = M, N, K, ta, tb, dt
    d.flags = flags | (L.EPI_OUT_F32 if out_f32 else 0)
    d.alpha = alpha; d.Z1 = d.Z2 = 1; d.splitk = splitk
    if bias is not None: d.bias = bias.data_ptr()
    if res is not None: d.residual = res.data_ptr()
    if aux is not None: d.aux = aux.data_ptr()
    d.drop_p, d.drop_seed = drop
    rc = h.vmr_gemm(C.byref(d), torch.cuda.current_stream().cuda_stream)
    assert rc == 0, h.vmr_last_error()
    return Cm

def mk(rows, cols, dt, lo=-3, hi=4):
    return torch.randint(lo, hi, (rows, cols), device=dev).to(dt)

bad = 0
for dt in (torch.bfloat16, torch.float32):
    for (M, N, K) in [(128, 128, 64), (256, 384, 192), (200, 136, 72), (70, 50, 33), (8, 8, 8), (1, 1, 5), (130, 4, 1024), (60, 96, 400)]:
        for ta in (0, 1):
            for tb in (0, 1):
                A = mk(K, M, dt) if ta else mk(M, K, dt)
                B = mk(K, N, dt) if tb else mk(N, K, dt)
                ref = (A.float().t() if ta else A.float()) @ (B.float() if tb else B.float().t())
                out = gemm(A, B, ta, tb, M, N, K, out_f32=True)
                torch.cuda.synchronize()
                err = (out - ref).abs().max().item()
                if err != 0:
                    bad += 1
                    print("MISMATCH", dt, M, N, K, ta, tb, err)
print("exact-integer checks done, bad =", bad)
# epilogue check
for dt in (torch.bfloat16, torch.float32):
    M, N, K = 192, 256, 128
    A = mk(M, K, dt); B = mk(N, K, dt); bias = torch.randn(N, device=dev); res = mk(M, N, dt)
    aux = torch.empty(M, N, device=dev, dtype=dt)
    out = gemm(A, B, 0, 0, M, N, K, flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_RESIDUAL | L.EPI_AUX, bias=bias, res=res, aux=aux)
    h_ref = torch.relu(A.float() @ B.float().t() + bias)
    ref = h_ref + res.float()
    print(dt, "epilogue err", (out.float() - ref).abs().max().item(), "aux err", (aux.float() - h_ref).abs().max().item())
    # split-K accumulate
    Cacc = torch.ones(M, N, device=dev)
    gemm(A, B, 0, 0, M, N, K, flags=L.EPI_ACCUM, splitk=2, Cacc=Cacc)
    print(dt, "splitk err", (Cacc - 1 - A.float() @ B.float().t()).abs().max().item())
    # dropout
    out = gemm(A, B, 0, 0, M, N, K, flags=L.EPI_DROPOUT, out_f32=True, drop=(0.25, 123))
    ref = A.float() @ B.float().t()
    kept = (out != 0) | (ref == 0)
    print(dt, "dropout keep frac", kept.float().mean().item(), "scaled err", ((out - ref / 0.75) * (out != 0)).abs().max().item())

# perf
def bench(M, N, K, ta, tb, dt=torch.bfloat16, iters=50):
    A = (torch.randn(K, M, device=dev) if ta else torch.randn(M, K, device=dev)).to(dt)
    B = (torch.randn(K, N, device=dev) if tb else torch.randn(N, K, device=dev)).to(dt)
    for _ in range(5): gemm(A, B, ta, tb, M, N, K)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): gemm(A, B, ta, tb, M, N, K)
    torch.cuda.synchronize(); dtm = (time.time() - t0) / iters
    # torch reference (hipBLASLt) for context only
    Am = A.t() if ta else A; Bm = B if tb else B.t()
    for _ in range(5): torch.matmul(Am, Bm)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): torch.matmul(Am, Bm)
    torch.cuda.synchronize(); dtt = (time.time() - t0) / iters
    print(f"{str(dt):16s} M{M} N{N} K{K} ta{ta} tb{tb}: {dtm*1e6:8.1f} us  {2*M*N*K/dtm/1e12:7.1f} TF | torch {dtt*1e6:8.1f} us {2*M*N*K/dtt/1e12:7.1f} TF")
for (M, N, K, ta, tb) in [(8192, 1024, 1024, 0, 0), (8192, 1024, 1024, 0, 1), (1024, 1024, 8192, 1, 1), (8192, 3072, 1024, 0, 0), (8192, 1024, 4096, 0, 0), (1280, 1024, 1024, 0, 0), (9472, 1024, 1024, 0, 0), (4096, 4096, 4096, 0, 0)]:
    bench(M, N, K, ta, tb)
bench(8192, 1024, 1024, 0, 0, torch.float32, 10)
bench(1024, 1024, 8192, 1, 1, torch.float32, 10)
