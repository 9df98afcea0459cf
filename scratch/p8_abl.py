import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
from vmrframe_amd import _lib as L
tag = "dbg=" + os.environ.get("VMR_P8_DBG", "0")
bench(4096, 4096, 4096, 0, 0, tag=tag)
bench(9472, 1024, 1024, 0, 0, tag=tag)
bench(1024, 1024, 9472, 1, 1, flags=L.EPI_SLAB, splitk=4, tag=tag)
