import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bdd_amd import native
from bdd_amd.solver import bdd_hip_parallel_mma
rng = np.random.Generator(np.random.PCG64(5))
k, B = 100, 20000; V = 2 * B
rows = [(np.ones(k, int), np.sort(rng.choice(V, size=k, replace=False)), ">=", 1) for _ in range(B)]
costs = rng.uniform(1, 10, V)
for split in (None, 50, 25):
    t = time.time(); col = native.rows_to_bdd_collection(rows, split_length=split, nr_variables=V)
    c = np.zeros(col.nr_variables()); c[:V] = costs
    s = bdd_hip_parallel_mma(col, c, precision="float")
    s.iterations(20); ms = s.time_iterations(500)
    lbs = []
    s2 = bdd_hip_parallel_mma(col, c, precision="double")
    for it in (100, 400, 1500):
        s2.iterations(it - (0 if not lbs else lbs[-1][0])); lbs.append((it, s2.lower_bound()))
    print("split", split, "bdds", col.nr_bdds(), "nodes", col.nr_bdd_nodes(), "hops", s.nr_hops(), "packs", s.nr_packs(), f"{500/ms*1e3:.0f} it/s", "LB@100/400/1500:", [round(x[1], 2) for x in lbs], f"build {time.time()-t:.1f}s")
