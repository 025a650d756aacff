"""Does the iteration rate of the headline instance depend on WHERE its arrays were allocated?  N solvers created one after another in one process
(every other one kept alive so that the next gets other addresses), each timed three times.
    python tools/placement_probe.py [N = 8] [precision = float] [V = 1000000]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])   # experimental builds under build/
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prec = sys.argv[2] if len(sys.argv) > 2 else "float"
V = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
col, costs = random_set_cover_mt(V, V // 2, 10, 12345)
IT = max(100, int(600 * 1_000_000 / V))
keep = []
for i in range(N):
    s = bdd_hip_parallel_mma(col, costs, precision=prec)
    s.iterations(200); s.synchronize()
    rates = []
    for _ in range(3):
        ms = s.time_iterations(IT)
        rates.append(IT / ms * 1e3)
    s.set_profiling(True, stride=8)
    s.iterations(640); s.synchronize()
    pr = s.get_profile()
    s.set_profiling(False)
    inloop = " ".join(f"{1e3 * t / max(n, 1):.1f}" for n, t in zip(pr["launches"], pr["total_ms"]) if n)
    print(f"solver {i}: " + " / ".join(f"{r:.0f}" for r in rates) + f" it/s   sweeps alone {s.time_kernel(2, 20) * 1e3:.1f} / {s.time_kernel(3, 20) * 1e3:.1f} us   in the loop (hipEvent pairs, us per launch by kernel class): {inloop}", flush=True)
    if i % 2 == 0:
        keep.append(s)
    else:
        s.close()
