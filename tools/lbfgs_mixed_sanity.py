import sys, time; sys.path.insert(0, ".")
import numpy as np
from bdd_amd import native
from bdd_amd.solver import bdd_hip_parallel_mma, bdd_hip_lbfgs
rng = np.random.Generator(np.random.PCG64(1)); V=200000; rows=[]
for _ in range(20000):
    vs = np.sort(rng.choice(V, size=14, replace=False)); co = rng.integers(1, 30, size=14); rows.append((co, vs, "<=", int(co.sum() // 2)))
for _ in range(250000):
    rows.append((np.ones(10, int), np.sort(rng.choice(V, size=10, replace=False)), ">=", 1))
col = native.rows_to_bdd_collection(rows); costs = rng.uniform(-10, 10, col.nr_variables())
for prec in ("float", "double"):
    s = bdd_hip_parallel_mma(col, costs, precision=prec); l = bdd_hip_lbfgs(s)
    prev = s.lower_bound(); t0 = time.perf_counter(); kinds = 0
    for i in range(60):
        l.iteration(); lb = l.lower_bound(); assert lb >= prev - 1e-5 * abs(prev), (i, lb, prev); prev = lb; kinds += l.state()["last_kind"]
    dt = time.perf_counter() - t0
    m = bdd_hip_parallel_mma(col, costs, precision=prec); m.iterations(60)
    print(prec, "lbfgs lb", round(prev, 3), "after 60 it in", round(dt * 1e3, 1), "ms,", kinds, "lbfgs steps; plain mma lb after 60 it", round(m.lower_bound(), 3))
