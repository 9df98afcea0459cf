"""Weight-gradient (TT, dz^T.x) products stand-alone: split-K sweep, slab epilogue, with VMR_GEMM_P8 from the environment."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
from vmrframe_amd import _lib as L
tag = "p8=" + os.environ.get("VMR_GEMM_P8", "1")
for sk in (1, 2, 4, 8, 16):
    bench(1024, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=sk, tag=tag)
for sk in (2, 4):
    bench(3072, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=sk, tag=tag)
bench(9472, 1024, 1024, 0, 0, tag=tag)
