"""L-BFGS iterations on the 10.5 M-node instance (or V variables, V / 2 rows) for a profiler run:  python tools/lbfgs_prof.py [float|double] [iters] [V] [variant_flags]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])   # experimental builds under build/
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_lbfgs, bdd_hip_parallel_mma
prec = sys.argv[1] if len(sys.argv) > 1 else "float"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
V = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
col, costs = random_set_cover_mt(V, V // 2, 10, 12345)
variant = int(sys.argv[4], 0) if len(sys.argv) > 4 else 0
s = bdd_hip_parallel_mma(col, costs, precision=prec, variant_flags=variant)
l = bdd_hip_lbfgs(s)
for _ in range(20):
    l.iteration()
s.lower_bound()
t0 = time.perf_counter()
trials = 0
for _ in range(iters):
    l.iteration()
    trials += l.state()["last_trials"]
lb = s.lower_bound()
dt = time.perf_counter() - t0
print(f"{prec}: {iters / dt:.1f} it/s, {1e6 * dt / iters:.1f} us per iteration, {trials / iters:.2f} trial steps per iteration, lb {lb:.6f}")
