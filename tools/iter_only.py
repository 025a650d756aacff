"""Plain iterations of one set-cover instance (for kernel-trace runs: per-kernel durations IN SEQUENCE, unlike kbench's back-to-back repeats)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="float")
ap.add_argument("--vars", type=int, default=1_000_000)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--iters", type=int, default=300)
ap.add_argument("--vars-per-bin", type=int, default=0)
ap.add_argument("--wpb", type=int, default=0)
a = ap.parse_args()
col, costs = random_set_cover_mt(a.vars, a.vars // 2, 10, seed=12345)
s = bdd_hip_parallel_mma(col, costs, precision=a.precision, variant_flags=a.variant, vars_per_bin=a.vars_per_bin, waves_per_block=a.wpb)
s.iterations(20)
ms = s.time_iterations(a.iters)
print(f"iteration = {ms / a.iters * 1e3:.1f} us ({a.iters / ms * 1e3:.0f} it/s) lb = {s.lower_bound():.9g}")
