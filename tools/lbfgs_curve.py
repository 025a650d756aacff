"""BASELINE.json configs[3]: L-BFGS around the HIP parallel-MMA backbone on the 10.5 M-node instance —
iterations/s and lower bound vs iteration / time, next to plain MMA.  Writes profiles/<tag>_lbfgs_curve.json.

    python tools/lbfgs_curve.py [--precision double] [--iters 150] [--tag r01]
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.instances import random_set_cover_mt as random_set_cover, set_cover_sizes
from bdd_amd.solver import bdd_hip_lbfgs, bdd_hip_parallel_mma

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="double")
ap.add_argument("--iters", type=int, default=150)
ap.add_argument("--vars", type=int, default=1_000_000)
ap.add_argument("--rows", type=int, default=500_000)
ap.add_argument("--tag", default="r02")
a = ap.parse_args()
col, costs = random_set_cover(a.vars, a.rows, 10, seed=12345)
out = {"workload": f"random set cover k=10, V={a.vars}, B={a.rows}: {set_cover_sizes(a.vars, a.rows, 10)['N']} BDD nodes",
       "precision": a.precision, "lbfgs": {"history size": 5, "initial step size": 1e-6, "required relative lb increase": 1e-6,
                                            "step size decrease factor": 0.8, "step size increase factor": 1.1}}
for name in ("mma", "lbfgs"):
    s = bdd_hip_parallel_mma(col, costs, precision=a.precision)
    stepper = bdd_hip_lbfgs(s) if name == "lbfgs" else s
    lbs, ts = [s.lower_bound()], [0.0]
    t0 = time.perf_counter()
    for _ in range(a.iters):
        stepper.iteration()
        lbs.append(s.lower_bound())      # run_solver reads the bound after every iteration (run_solver_util.h:37-41)
        ts.append(time.perf_counter() - t0)
    out[name] = {"iterations_per_s": a.iters / ts[-1], "lower_bound": lbs, "seconds": ts}
    print(name, "it/s", round(a.iters / ts[-1], 1), "lb[10,50,last]", lbs[10], lbs[50], lbs[-1])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(f"gpurun_out/{a.tag}_lbfgs_curve_{a.precision}.json", "w"))
