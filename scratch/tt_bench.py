import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
from vmrframe_amd import _lib as L
tag = "dma=" + os.environ.get("VMR_GEMM_DMA", "2")
for sk in (4, 8):
    bench(1024, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=sk, tag=tag)
bench(3072, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=2, tag=tag)
bench(3072, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=4, tag=tag)
bench(1024, 512, 8192, 1, 1, flags=L.EPI_SLAB, splitk=8, tag=tag)
bench(9472, 1024, 1024, 0, 0, tag=tag)
