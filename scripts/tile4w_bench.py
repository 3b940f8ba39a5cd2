"""Experiment: 256 x 256 NT tile on FOUR waves (128 x 128 per wave: 16 fragment reads per 64 MFMAs instead of 12 per 32) against
the 8-wave tile, plain GEMMs of the hot shapes.  Needs a library built with -DEVC_EXPERIMENT_4WAVE (EVC_LIB); the tile is chosen by
EVC_FORCE_TILE (12: 4 waves, 1: 8 waves), read once per process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gemm_bench import gemm
t = os.environ.get("EVC_FORCE_TILE", "0")
for (M, N, K) in ((3840, 4096, 2176), (4096, 4096, 4096), (8192, 8192, 8192), (56640, 1024, 4096), (5120, 4096, 4096)):
    gemm(M, N, K, "tile %s" % t)
