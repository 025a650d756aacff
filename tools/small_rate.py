"""Iteration rate of instances that fit one workgroup, fused (k_iterate_small) vs four launches per iteration (variant_flags bit 19):
   python tools/small_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import to_bdd_collection
from bdd_amd.instances import assignment_ilp, random_set_cover
from bdd_amd.solver import bdd_hip_parallel_mma, run_solver


def rate(s, n):
    s.iterations(200); s.synchronize()
    t0 = time.perf_counter(); s.iterations(n); s.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


cases = [("8 x 8 assignment", lambda: (to_bdd_collection(assignment_ilp(8)), assignment_ilp(8).objective)),
         ("3 x 3 assignment", lambda: (to_bdd_collection(assignment_ilp(3)), assignment_ilp(3).objective)),
         ("20 x 20 assignment", lambda: (to_bdd_collection(assignment_ilp(20)), assignment_ilp(20).objective)),
         ("set cover 60 rows k=5", lambda: random_set_cover(40, 60, 5, seed=60)),
         ("set cover 220 rows k=8", lambda: random_set_cover(146, 220, 8, seed=220)),
         ("set cover 300 rows k=9", lambda: random_set_cover(200, 300, 9, seed=300)),
         ("set cover 500 rows k=12", lambda: random_set_cover(333, 500, 12, seed=500))]
for name, make in cases:
    col, costs = make()
    for prec in ("float", "double"):
        f = bdd_hip_parallel_mma(col, costs, precision=prec)
        q = bdd_hip_parallel_mma(col, costs, precision=prec, variant_flags=0x80000)
        uf, uq = rate(f, 20000 if f.fused_small() else 2000), rate(q, 2000)
        rf = run_solver(f, max_iter=2000, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)
        rq = run_solver(q, max_iter=2000, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)
        print(f"{name:26s} {prec:6s} packs {f.nr_packs():3d} fused {int(f.fused_small())}: iterations(n) {uf:7.2f} us/it (four launches {uq:6.2f});  "
              f"run_solver {rf['seconds'] / rf['iterations'] * 1e6:7.2f} us/it (four launches {rq['seconds'] / rq['iterations'] * 1e6:6.2f})", flush=True)
