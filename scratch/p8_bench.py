import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
from vmrframe_amd import _lib as L
tag = "p8=" + os.environ.get("VMR_GEMM_P8", "0")
for (M, N, K) in [(9472, 1024, 1024), (8192, 1024, 1024), (9472, 3072, 1024), (9472, 2048, 1024), (8192, 1024, 4096), (4096, 4096, 4096), (8192, 8192, 8192)]:
    bench(M, N, K, 0, 0, tag=tag)
bench(9472, 1024, 1024, 0, 0, epi=True, tag=tag)
for sk in (4, 6, 8):
    bench(1024, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=sk, tag=tag)
bench(3072, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=2, tag=tag)
