"""What the backward solve sweep pays for writing x in layer order (L-BFGS wrapper attached): plain iterations before / after.
python tools/xlayer_cost.py [float|double]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_lbfgs, bdd_hip_parallel_mma
prec = sys.argv[1] if len(sys.argv) > 1 else "float"
col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
s = bdd_hip_parallel_mma(col, costs, precision=prec)
s.iterations(20)
names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exch_reduce", "exch_bcast"]
def show(tag):
    ms = s.time_iterations(400)
    print(tag, "  ".join(f"{n}={s.time_kernel(k, 30)*1e3:.1f}us" for k, n in enumerate(names)), f"  iteration = {ms / 400 * 1e3:.1f} us")
show("before:")
l = bdd_hip_lbfgs(s)
l.iteration()
show("after: ")
show("after: ")
