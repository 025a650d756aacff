"""What bounds the sweeps once nothing is cached?  python tools/hbm_diag.py [--vars V] [--precision float]: a few launches of the four
sweep kernels, the exchange and the STREAM probes of one instance, to be run under rocprofv3 --pmc (tools/hbm_diag.sh)."""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
ap = argparse.ArgumentParser()
ap.add_argument("--vars", type=int, default=4_000_000)
ap.add_argument("--precision", default="float")
ap.add_argument("--wpb", type=int, default=0)
ap.add_argument("--reps", type=int, default=10)
a = ap.parse_args()
col, costs = random_set_cover_mt(a.vars, a.vars // 2, 10, seed=12345)
s = bdd_hip_parallel_mma(col, costs, precision=a.precision, waves_per_block=a.wpb)
s.iterations(3)
names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exch_reduce", "exch_bcast", "triad", "copy"]
print("  ".join(f"{n}={s.time_kernel(k, a.reps)*1e3:.1f}us" for k, n in enumerate(names)))
