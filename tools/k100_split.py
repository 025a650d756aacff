"""Row size k = 100 at the headline size (52 238 covering rows of 100 variables, ~10.5 M nodes): what the JSON key "split bdds" buys.
Unsplit / cut at 50 / cut at 25 layers (split_qbdd with auxiliary one-hot variables, bdd_collection.cpp:507-949)."""
import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bdd_amd import native
from bdd_amd.solver import bdd_hip_parallel_mma
rng = np.random.Generator(np.random.PCG64(5))
k, B = 100, 52238; V = 2 * B
rows = [(np.ones(k, int), np.sort(rng.choice(V, size=k, replace=False)), ">=", 1) for _ in range(B)]
costs = rng.uniform(1, 10, V)
for split in (None, 50, 25):
    t = time.time(); col = native.rows_to_bdd_collection(rows, split_length=split, nr_variables=V)
    c = np.zeros(col.nr_variables()); c[:V] = costs
    out = [f"split {split}: {col.nr_bdds()} BDDs, {col.nr_bdd_nodes()} nodes"]
    for prec in ("float", "double"):
        s = bdd_hip_parallel_mma(col, c, precision=prec)
        s.iterations(20); ms = s.time_iterations(300)
        out.append(f"{prec} {300 / ms * 1e3:.0f} it/s ({s.nr_hops()} hops, {s.nr_packs()} packs)")
        if prec == "double":
            s.iterations(180)
            out.append(f"LB after 500 iterations {s.lower_bound():.3f}")
        s.close()
    print("; ".join(out), f"; build {time.time() - t:.1f} s", flush=True)
