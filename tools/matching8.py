"""8 x 8 assignment problem (BASELINE.json configs[0] on the GPU): pure dependent latency, one pack, one exchange bin.  python tools/matching8.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bdd_amd.instances import assignment_ilp
from bdd_amd import to_bdd_collection
from bdd_amd.solver import bdd_hip_parallel_mma
ilp = assignment_ilp(8)
col, costs = to_bdd_collection(ilp), np.asarray(ilp.objective, float)
for prec in ("double", "float"):
    s = bdd_hip_parallel_mma(col, costs, precision=prec)
    s.iterations(100)
    ms = s.time_iterations(5000) / 5000
    print(prec, f"{ms * 1e3:.2f} us per iteration", [round(s.time_kernel(k, 50) * 1e3, 2) for k in (2, 3, 4)])
