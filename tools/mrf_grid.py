import os, sys, time; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from bdd_amd import capi, native
from bdd_amd.instances import mrf_ilp
from bdd_amd.solver import bdd_hip_parallel_mma, bdd_hip_lbfgs, run_solver
n = int(sys.argv[1])
rng = np.random.Generator(np.random.PCG64(3))
t = time.time()
un = [tuple(rng.normal(0, 1, 2).round(3)) for _ in range(n * n)]
edges = []
for i in range(n):
    for j in range(n):
        u = i * n + j
        if j + 1 < n: edges.append((u, u + 1, tuple(rng.normal(0, 2, 4).round(3))))
        if i + 1 < n: edges.append((u, u + n, tuple(rng.normal(0, 2, 4).round(3))))
ilp = mrf_ilp(un, edges)
print("ilp", ilp.nr_variables(), len(ilp.constraints), time.time() - t); t = time.time()
col = native.rows_to_bdd_collection([(c.coefficients, c.variables, c.ineq, c.rhs) for c in ilp.constraints])
print("bdds", col.nr_bdds(), "nodes", col.nr_bdd_nodes(), time.time() - t)
for prec in ("float", "double"):
    t = time.time(); s = bdd_hip_parallel_mma(col, ilp.objective, precision=prec); print(prec, "create", time.time() - t, "packs", s.nr_packs(), "hops", s.nr_hops())
    r = run_solver(s, 2000, 1e-7, 1e-9, 60, verbose=False); print(" mma", r)
    ms = s.time_iterations(500); print(f" {500/ms*1e3:.0f} it/s")
    L = capi.lib(); sol = np.zeros(s.nr_variables(), np.int8); found = C.c_int(0)
    t = time.time(); L.bddmma_incremental_mm_agreement_rounding(s._h, None, 0.1, 1.1, 100, 100, 0, 0, sol.ctypes.data_as(C.c_void_p), C.byref(found))
    x = sol[: ilp.nr_variables()].tolist()
    print(" rounding found", found.value, "time", round(time.time() - t, 2), "feasible", ilp.feasible(x) if found.value else None, "obj", ilp.evaluate(x) if found.value else None, "lb", r["lb_final"])
s = bdd_hip_parallel_mma(col, ilp.objective, precision="double"); lb = bdd_hip_lbfgs(s)
r = run_solver(s, 2000, 1e-7, 1e-9, 60, verbose=False, lbfgs=lb); print("lbfgs", r)
