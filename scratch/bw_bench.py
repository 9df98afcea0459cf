import torch
dev="cuda"
def bench(fn, nbytes, name, iters=24):
    for i in range(8): fn(i)
    torch.cuda.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): fn(i)
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)*1e-3/iters
    print(f"{name}: {t*1e6:.1f} us  {nbytes/t/1e12:.2f} TB/s")
for mb in (19.4, 38.8, 77.6, 310):
    n=int(mb*1e6/2)
    nset=max(2,int(1200/mb)) if mb<200 else 4
    xs=[torch.randn(n,device=dev).bfloat16() for _ in range(nset)]
    ys=[torch.empty_like(x) for x in xs]
    bench(lambda i: ys[i%nset].copy_(xs[i%nset]), 2*n*2, f"copy {mb}MB src (cold, {nset} sets)")
    bench(lambda i: ys[0].copy_(xs[0]), 2*n*2, f"copy {mb}MB src (hot)")
    bench(lambda i: xs[i%nset].sum(), n*2, f"sum-read {mb}MB (cold)")
    bench(lambda i: ys[i%nset].fill_(1.0), n*2, f"fill-write {mb}MB (cold)")
