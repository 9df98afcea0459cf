"""8-phase 256x256 GEMM kernel: exactness on integer operands (run with VMR_GEMM_P8=2), then timings."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import ops, _lib as L
dev = "cuda"; dt = torch.bfloat16
def ints(r, c): return torch.randint(-3, 4, (r, c), device=dev).to(dt)
ok = True
for (M, N, K) in [(256, 256, 128), (512, 256, 192), (768, 512, 64 * 7), (9472, 1024, 1024), (1288, 256, 256), (264, 512, 128), (9472, 3072, 320)]:
    torch.manual_seed(M + N + K)
    A, B = ints(M, K), ints(N, K)
    ref = A.float() @ B.float().t()
    out = ops.mm(A, B, 0, 0, out_f32=True)
    e = torch.equal(out, ref); ok &= e
    print("NT", M, N, K, "exact" if e else f"MISMATCH max {float((out-ref).abs().max())} bad {int((out!=ref).sum())}", flush=True)
    o16 = ops.mm(A, B, 0, 0)
    e = torch.equal(o16, ref.to(dt)); ok &= e
    print("NT bf16 out", "exact" if e else "MISMATCH", flush=True)
    # full epilogue
    bias = torch.randn(N, device=dev); res = ints(M, N)
    out = torch.empty(M, N, device=dev, dtype=dt); aux = torch.empty_like(out)
    ops.gemm(A, B, out, M, N, K, 0, 0, K, K, N, dtype=L.BF16, bias=bias, residual=res, aux=aux, ldr=N,
             flags=L.EPI_BIAS | L.EPI_RELU | L.EPI_DROPOUT | L.EPI_RESIDUAL | L.EPI_AUX, drop=(0.25, 9, None))
    mask = ops.dropout_mask(M * N, 0.25, 9, dev).view(M, N)
    h = torch.relu(ref + bias) * mask
    e = torch.equal(aux, h.to(dt)) and torch.equal(out, (h + res.float()).to(dt)); ok &= e
    print("NT epilogue", "exact" if e else "MISMATCH", flush=True)
for (M, N, K, sk) in [(256, 256, 256, 1), (1024, 1024, 2368, 4), (1024, 1024, 9472, 6), (3072, 1024, 9472, 2), (512, 256, 64 * 9, 3)]:
    torch.manual_seed(K)
    A, B = ints(K, M), ints(K, N)
    ref = A.float().t() @ B.float()
    if sk == 1:
        out = ops.mm(A, B, 1, 1, out_f32=True)
    else:
        ws = torch.empty(sk, M, N, device=dev)
        ops.gemm(A, B, ws, M, N, K, 1, 1, M, N, N, dtype=L.BF16, flags=L.EPI_SLAB, splitk=sk)
        out = ws.sum(0)
    e = torch.equal(out, ref); ok &= e
    print("TT", M, N, K, sk, "exact" if e else f"MISMATCH max {float((out-ref).abs().max())} bad {int((out!=ref).sum())}", flush=True)
print("ALL OK" if ok else "FAILURES")
