"""Timing experiment (wrong results by design): iteration time of the fused small-instance kernel with phases skipped (build/libsmall_skip<mask>.so,
BDDMMA_EXP_SMALL_SKIP bit 0 forward sweep, 1 first exchange, 2 backward sweep, 3 second exchange).  python tools/small_phases.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])
from bdd_amd import to_bdd_collection
from bdd_amd.instances import assignment_ilp, random_set_cover
from bdd_amd.solver import bdd_hip_parallel_mma
for name, (col, costs) in (("8x8", (to_bdd_collection(assignment_ilp(8)), assignment_ilp(8).objective)), ("cover220", random_set_cover(146, 220, 8, seed=220))):
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    s.iterations(2000); s.synchronize()
    t0 = time.perf_counter(); s.iterations(50000); s.synchronize()
    print(f"{os.environ.get('BDDMMA_LIB', 'shipped'):28s} {name:9s} {(time.perf_counter() - t0) / 50000 * 1e6:6.2f} us per iteration", flush=True)
