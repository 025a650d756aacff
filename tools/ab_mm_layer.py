import os, sys, time
sys.path.insert(0, os.getcwd())
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_lbfgs, bdd_hip_parallel_mma
for prec in ("float", "double"):
    col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
    s = bdd_hip_parallel_mma(col, costs, precision=prec)
    s.iterations(500); s.lower_bound()
    def t(n=2000):
        s.lower_bound(); t0 = time.perf_counter(); s.iterations(n); s.lower_bound(); return 1e6 * (time.perf_counter() - t0) / n
    a = [t() for _ in range(3)]
    l = bdd_hip_lbfgs(s)
    for _ in range(3): l.iteration()
    b = [t() for _ in range(3)]
    print(prec, "us/iteration without mm_layer", [round(x, 1) for x in a], "with", [round(x, 1) for x in b])
