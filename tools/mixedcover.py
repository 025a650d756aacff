"""Set cover with row sizes drawn from [k_min, k_max] (instances.random_set_cover_mixed): BDDs of k_max - k_min + 1 lengths in one launch.
    python tools/mixedcover.py [--vars 1000000] [--rows 560000] [--kmin 3] [--kmax 16] [--keep-order 0] [--precision float,double]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mixed
from bdd_amd.solver import bdd_hip_parallel_mma
ap = argparse.ArgumentParser()
ap.add_argument("--vars", type=int, default=1_000_000)
ap.add_argument("--rows", type=int, default=560_000)
ap.add_argument("--kmin", type=int, default=3)
ap.add_argument("--kmax", type=int, default=16)
ap.add_argument("--keep-order", type=int, default=0, dest="keep")
ap.add_argument("--precision", default="float,double")
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--wpb", type=int, default=0)
ap.add_argument("--pack-width", type=int, default=0)
ap.add_argument("--variant", type=int, default=0)
a = ap.parse_args()
col, costs = random_set_cover_mixed(a.vars, a.rows, a.kmin, a.kmax)
print(f"{col.nr_bdds()} BDDs, {col.nr_bdd_nodes()} nodes, keep_bdd_order {a.keep}")
for prec in a.precision.split(","):
    s = bdd_hip_parallel_mma(col, costs, precision=prec, keep_bdd_order=a.keep, waves_per_block=a.wpb, pack_width=a.pack_width, variant_flags=a.variant)
    s.iterations(5)
    names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exchange"]
    print(f"{prec}: packs {s.nr_packs()} sweeps {s.solve_sweep_kind()} " + "  ".join(f"{n}={s.time_kernel(i, 20) * 1e3:.1f}us" for i, n in enumerate(names)))
    ms = s.time_iterations(a.iters) / a.iters
    print(f"   iteration {ms * 1e3:.1f} us = {1e3 / ms:.0f} it/s   lb {s.lower_bound():.6f}")
