import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_bench import gemm
gemm(4096, 4096, 4096, "square 4096 " + os.environ.get("EVC_LIB", "")[-20:])
gemm(8192, 8192, 8192, "square 8192")
