import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gemm_bench import bench
tag = "big=" + os.environ.get("VMR_GEMM_BIG", "1")
for (M, N, K) in [(8192, 1024, 1024), (9472, 1024, 1024), (9472, 3072, 1024), (9472, 2048, 1024), (8192, 1024, 4096), (8192, 3072, 1024), (1280, 1024, 1024)]:
    bench(M, N, K, 0, 0, tag=tag)
bench(9472, 1024, 1024, 0, 0, epi=True, tag=tag)
bench(8192, 1024, 1024, 0, 0, epi=True, tag=tag)
