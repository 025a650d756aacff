"""Stress of one run_solver case under load: python tests/tools/stress_run_solver.py LOOPS [precision]
Each loop builds a twin pair (deterministic exchange), runs the reference's sequential loop on one and bddmma_run_solver on the other and prints
everything on a mismatch (iteration counts, reasons, the bounds of the last iterations, NaN counts of the solver costs)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from bdd_amd.instances import random_set_cover
from bdd_amd.solver import bdd_hip_parallel_mma, run_solver

loops = int(sys.argv[1]) if len(sys.argv) > 1 else 20
precision = sys.argv[2] if len(sys.argv) > 2 else "double"
col, costs = random_set_cover(3000, 2500, 8, seed=21)
bad = 0
for it in range(loops):
    twin = bdd_hip_parallel_mma(col, costs, precision=precision, deterministic=True)
    lbs = [twin.lower_bound()]
    lb_initial = lbs[0]; lb_first = float(np.finfo(np.float64).max); reason = 0; n = 400
    for k in range(400):
        twin.iteration(); lbs.append(twin.lower_bound())
        if k == 0: lb_first = lbs[-1]
        if abs(lbs[-2] - lbs[-1]) < abs(1e-6 * lbs[-2]): reason = 2
        elif abs(lbs[-2] - lbs[-1]) < 1e-9 * abs(lb_initial - lb_first): reason = 3
        elif lbs[-1] == math.inf: reason = 4
        if reason: n = k + 1; break
    s = bdd_hip_parallel_mma(col, costs, precision=precision, deterministic=True)
    res = run_solver(s, max_iter=400, tolerance=1e-6, improvement_slope=1e-9, time_limit=1e9)
    ok = (res["iterations"], res["stop_reason"]) == (n, reason) and res["lb_final"] == lbs[-1]
    if not ok:
        bad += 1
        cs = [np.asarray(a) for a in s.get_solver_costs()]; ct = [np.asarray(a) for a in twin.get_solver_costs()]
        print(f"MISMATCH loop {it}: run_solver {res['iterations']} reason {res['stop_reason']} lb {res['lb_final']!r} initial {res['lb_initial']!r} | twin {n} reason {reason} "
              f"lbs[:3] {lbs[:3]} lbs[-3:] {lbs[-3:]} nan(twin lbs) {sum(1 for x in lbs if x != x)} | nan costs s {[int(np.isnan(a).sum()) for a in cs]} twin {[int(np.isnan(a).sum()) for a in ct]}", flush=True)
print(f"pid {os.getpid()}: {bad} mismatches in {loops} loops", flush=True)
