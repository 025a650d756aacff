"""Wide-pack path micro-benchmark: knapsack rows (layer width up to ~150 nodes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bdd_amd import BddCollection
from bdd_amd.solver import bdd_hip_parallel_mma
from oracle.oracle import Oracle
rng = np.random.Generator(np.random.PCG64(1))
V, B, k = 20000, 4000, 14
col = BddCollection()
t0 = time.time()
for _ in range(B):
    vs = np.sort(rng.choice(V, size=k, replace=False))
    co = rng.integers(1, 30, size=k)
    col.add_linear(co, "<=", int(co.sum() // 2), vs)
idx = col.instr[:, 2]; idx = idx[idx < 2**63]
print("built", col.nr_bdds(), "BDDs,", col.nr_bdd_nodes(), "nodes in", round(time.time() - t0, 1), "s")
costs = -rng.uniform(1, 10, col.nr_variables())
pw = int(sys.argv[1]) if len(sys.argv) > 1 else 0
variant = int(sys.argv[2]) if len(sys.argv) > 2 else 0
wpb = int(sys.argv[3]) if len(sys.argv) > 3 else 0
vpb = int(sys.argv[4]) if len(sys.argv) > 4 else 0
for prec in ("float", "double"):
    s = bdd_hip_parallel_mma(col, costs, precision=prec, pack_width=pw, variant_flags=variant, waves_per_block=wpb, vars_per_bin=vpb)
    o = Oracle(col, costs, prec, threads=16)
    for _ in range(5):
        s.iteration(); o.iteration()
    print(prec, "packs", s.nr_packs(), "lb", s.lower_bound(), o.lower_bound())
    names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exchange"]
    print("  ".join(f"{n}={s.time_kernel(i, 20)*1e3:.1f}us" for i, n in enumerate(names)))
    print("it/s", round(200 / (s.time_iterations(200) * 1e-3)))
