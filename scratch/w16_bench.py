"""VMR_GEMM_W16 experiment: x.W^T products of the step with the 8-wave 128x128 tile (16 waves per CU) vs the default tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
tag = "w16=" + os.environ.get("VMR_GEMM_W16", "0")
for (M, N, K) in [(8192, 1024, 1024), (9472, 1024, 1024), (8192, 1024, 4096), (9472, 3072, 1024)]:
    bench(M, N, K, 0, 0, tag=tag)
bench(8192, 1024, 1024, 0, 0, epi=True, tag=tag)
