import sys, os; sys.path.insert(0,".")
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma, run_solver
v = int(sys.argv[1]) if len(sys.argv) > 1 else 0
col,costs=random_set_cover_mt(1000000,500000,10,12345)
s=bdd_hip_parallel_mma(col,costs,precision="float", variant_flags=v)
s.iterations(300)
run_solver(s,max_iter=2,tolerance=0.0,improvement_slope=0.0,time_limit=1e9)
for _ in range(3):
    r=run_solver(s,max_iter=2000,tolerance=0.0,improvement_slope=0.0,time_limit=1e9)
    print("variant", v, "run_solver it/s", round(r["iterations"]/r["seconds"]), end="; ")
import time
t0=time.perf_counter(); s.iterations(2000); s.lower_bound(); print("plain", round(2000/(time.perf_counter()-t0)))
