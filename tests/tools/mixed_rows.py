import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from bdd_amd.instances import random_set_cover_mixed, random_set_cover
from bdd_amd.solver import bdd_hip_parallel_mma
from oracle.oracle import Oracle
col, costs = random_set_cover_mixed(1_000_000, 500_000, 3, 16, seed=7)
print("nodes", col.nr_bdd_nodes())
names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exch_reduce", "exch_bcast"]
for prec in ("float", "double"):
  for keep in (True, False):
    s = bdd_hip_parallel_mma(col, costs, precision=prec, keep_bdd_order=keep)
    s.iterations(3)
    t = "  ".join(f"{n}={s.time_kernel(k, 30)*1e3:.1f}us" for k, n in enumerate(names))
    n = 200; ms = s.time_iterations(n)
    print(prec, "keep_order" if keep else "grouped", "packs", s.nr_packs(), t, f"iteration={ms/n*1e3:.1f}us {n/ms*1e3:.0f} it/s lb={s.lower_bound():.9g}")
