"""run_solver (bound + termination tests every iteration, on the device) vs the plain loop:  python tools/run_solver_rate.py [variant_flags] [V]"""
import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma, run_solver
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
V = int(sys.argv[2]) if len(sys.argv) > 2 else 1000000
col,costs=random_set_cover_mt(V,V//2,10,12345)
s=bdd_hip_parallel_mma(col,costs,precision="float", variant_flags=v)
s.iterations(300)
run_solver(s,max_iter=2,tolerance=0.0,improvement_slope=0.0,time_limit=1e9)
for _ in range(3):
    r=run_solver(s,max_iter=2000,tolerance=0.0,improvement_slope=0.0,time_limit=1e9)
    print("variant", v, "run_solver it/s", round(r["iterations"]/r["seconds"]), end="; ")
import time
t0=time.perf_counter(); s.iterations(2000); s.lower_bound(); print("plain", round(2000/(time.perf_counter()-t0)))
