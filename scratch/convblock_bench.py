"""Row half of one conv-block layer's backward at cfg2 shapes (9472 x 1024, 64 clips x 128 frames + 64 sentences x 20
words): the three launches (dwconv_bwd2 + layernorm_bwd_deferred + relu_bwd_bias mode 3) against the one fused launch
(vmr_convblock_bwd).  Graph-timed, 20 launches per replay.  VMR_CONVBLOCK_GRID overrides the persistent grid (default: one workgroup per CU)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vmrframe_amd import _lib as L

dev = "cuda"
dt = torch.float16 if "--fp16" in sys.argv else torch.bfloat16
code = L.dtype_code(torch.empty(0, dtype=dt))


def timeit(name, fn, bytes_, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    print(f"{name:44s} {t * 1e6:8.1f} us  {bytes_ / t / 1e12:6.2f} TB/s (algorithmic {bytes_ / 1e6:.1f} MB)", flush=True)
    return t


B1, S1, B2, S2, D = 64, 128, 64, 20, 1024
N = B1 * S1 + B2 * S2
lib = L.lib()
st = lambda: torch.cuda.current_stream().cuda_stream
x = torch.randn(N, D, device=dev).to(dt); du = torch.randn(N, D, device=dev).to(dt); dres = torch.randn(N, D, device=dev).to(dt)
dn = torch.empty_like(x); dx = torch.empty_like(x); dz = torch.empty_like(x)
g = torch.ones(D, device=dev); b = torch.zeros(D, device=dev)
mean = x.float().mean(1); rstd = torch.rsqrt(x.float().var(1, unbiased=False) + 1e-6)
w = torch.randn(D, 7, device=dev)
bits = torch.randint(0, 256, (N, D // 8), device=dev, dtype=torch.uint8)
ws = torch.empty((B1 * 2 + B2) * D * 7, device=dev); ws2 = torch.empty(L.ln_bwd_ws_floats(N, D), device=dev)
nb = C.c_int32(0)


def three():
    lib.vmr_dwconv_bwd2_deferred(du.data_ptr(), x.data_ptr(), g.data_ptr(), b.data_ptr(), mean.data_ptr(), rstd.data_ptr(), w.data_ptr(),
                                 dn.data_ptr(), ws.data_ptr(), B1, S1, B2, S2, D, code, C.byref(nb), st())
    lib.vmr_layernorm_bwd_deferred(dn.data_ptr(), x.data_ptr(), g.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dres.data_ptr(), dx.data_ptr(),
                                   None, ws2.data_ptr(), 0, N, D, code, 0.0, 0, None, C.byref(nb), st())
    lib.vmr_relu_bwd_bias(3, dx.data_ptr(), bits.data_ptr(), dz.data_ptr(), None, N, D, D, 1.25, code, 0.0, 0, None, None, 1.0, st())


nbm = lib.vmr_convblock_bwd_blocks(B1, S1, B2, S2, D)
pdw = torch.empty(nbm, 7 * D, device=dev); pgb = torch.empty(nbm, 2 * D, device=dev)


def fused():
    rc = lib.vmr_convblock_bwd(du.data_ptr(), x.data_ptr(), dres.data_ptr(), bits.data_ptr(), 1.25, g.data_ptr(), b.data_ptr(), mean.data_ptr(),
                               rstd.data_ptr(), w.data_ptr(), dx.data_ptr(), dz.data_ptr(), pdw.data_ptr(), pgb.data_ptr(), B1, S1, B2, S2, D,
                               code, C.byref(nb), st())
    assert rc == 0, lib.vmr_last_error()


def fused_nodz():
    lib.vmr_convblock_bwd(du.data_ptr(), x.data_ptr(), dres.data_ptr(), None, 1.0, g.data_ptr(), b.data_ptr(), mean.data_ptr(),
                          rstd.data_ptr(), w.data_ptr(), dx.data_ptr(), None, pdw.data_ptr(), pgb.data_ptr(), B1, S1, B2, S2, D,
                          code, C.byref(nb), st())


e = N * D * 2
print(f"workgroups of the fused launch: {nbm} (VMR_CONVBLOCK_GRID={os.environ.get('VMR_CONVBLOCK_GRID', 'default')})")
t3 = timeit("dwconv_bwd2 + ln_bwd + relu_bwd_bias(3)", three, 9 * e)
t1 = timeit("vmr_convblock_bwd (dx + dz)", fused, 5 * e + N * D // 8)
timeit("vmr_convblock_bwd (dx only: bottom layer)", fused_nodz, 4 * e)
print(f"per layer: {t3 * 1e6:.1f} -> {t1 * 1e6:.1f} us")
