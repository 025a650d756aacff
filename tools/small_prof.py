"""rocprofv3 target: the 8 x 8 assignment problem, four launches of 16 384 iterations each per precision (k_iterate_small) — the kernel's average duration / 16 384
is the per-iteration time without any host clock:   bash tools/kstats.sh r06_small tools/small_prof.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import to_bdd_collection
from bdd_amd.instances import assignment_ilp
from bdd_amd.solver import bdd_hip_parallel_mma

ilp = assignment_ilp(8)
for prec in ("float", "double"):
    s = bdd_hip_parallel_mma(to_bdd_collection(ilp), ilp.objective, precision=prec)
    assert s.fused_small()
    for _ in range(4):
        s.iterations(16384)
    s.synchronize()
    print(prec, "lower bound", s.lower_bound())
