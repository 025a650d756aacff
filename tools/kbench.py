"""Kernel-level micro-benchmark on the GPU box: python tools/kbench.py [--precision float] [--pack-width N] ..."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])   # experimental builds under build/
from bdd_amd.instances import random_set_cover, random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="float")
ap.add_argument("--pack-width", type=int, default=0)
ap.add_argument("--vars-per-bin", type=int, default=0)
ap.add_argument("--stage-cap", type=int, default=0)
ap.add_argument("--wpb", type=int, default=0)
ap.add_argument("--vars", type=int, default=1_000_000)
ap.add_argument("--rows", type=int, default=500_000)
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--res", type=int, default=0, help="resident sweeps: 0 auto, 1 off, 2 on")
ap.add_argument("--byvar", type=int, default=0, help="entries by (variable, bdd): 0 auto, 1 off, 2 on")
ap.add_argument("--fill", type=int, default=0, help="bddmma_options.pack_fill")
ap.add_argument("--variant", type=int, default=0, help="bddmma_options.variant_flags")
ap.add_argument("--mt", type=int, default=0, help="1: the bench.py instance (std::mt19937_64 draw order of csrc/host/instances.cpp)")
ap.add_argument("--iters", type=int, default=200)
a = ap.parse_args()
col, costs = (random_set_cover_mt if a.mt else random_set_cover)(a.vars, a.rows, a.k, seed=12345)
s = bdd_hip_parallel_mma(col, costs, precision=a.precision, pack_width=a.pack_width, vars_per_bin=a.vars_per_bin, stage_cap=a.stage_cap, waves_per_block=a.wpb, resident_sweeps=a.res, exchange_by_variable=a.byvar, variant_flags=a.variant, pack_fill=a.fill)
s.iterations(3)
names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exch_reduce", "exch_bcast"]
print(vars(a))
print("  ".join(f"{n}={s.time_kernel(k, 30)*1e3:.1f}us" for k, n in enumerate(names)))
n = a.iters
ms = s.time_iterations(n)
print(f"  iteration = {ms / n * 1e3:.1f} us  ({n / ms * 1e3:.0f} it/s)   lb = {s.lower_bound():.9g}")
