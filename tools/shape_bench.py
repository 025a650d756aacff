"""One line per (shape, precision): sizes, the sweep family the rules chose, iterations/s and the SURVEY §8(d) whole-iteration roofline fraction
(B_iter = 2 [12 N' + 2R N + (5R+4) L' + (8R+4) V]; N incl. two terminals per BDD).  Shapes:
    cover K        random set cover, rows of K variables, ~10.5 M nodes (SURVEY §8d's k-sweep)
    assign N [S]   N x N assignment problem (2N simplex BDDs of N hops); S: "split bdds" with split length S (0: the occupancy rule)
python tools/shape_bench.py cover 4 | assign 1000 | assign 1000 0"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])   # experimental builds under build/
from bdd_amd import native
from bdd_amd.instances import random_set_cover_mt
from bdd_amd.solver import bdd_hip_parallel_mma

kind = sys.argv[1]
t0 = time.time()
if kind == "cover":
    k = int(sys.argv[2])
    B = 10_500_000 // (2 * k + 1)
    V = 2 * B
    col, costs = random_set_cover_mt(V, B, k, seed=12345)
    name = f"set cover k={k} B={B} V={V}"
elif kind == "assign":
    n = int(sys.argv[2])
    split = int(sys.argv[3]) if len(sys.argv) > 3 else None
    ones = np.ones(n, int)
    rows = [(ones, np.arange(i * n, (i + 1) * n), "=", 1) for i in range(n)] + [(ones, np.arange(j, n * n, n), "=", 1) for j in range(n)]
    col = native.rows_to_bdd_collection(rows, split_length=split, nr_variables=n * n)
    costs = np.concatenate([(-np.ones((n, n)) - np.eye(n)).ravel(), np.zeros(max(0, col.nr_variables() - n * n))])
    name = f"assignment {n} x {n}" + ("" if split is None else f", split bdds (length {split or 'auto'})")
else:
    raise SystemExit("unknown shape")
N = col.nr_bdd_nodes()
Bn = col.nr_bdds()
build = time.time() - t0
for prec, R in (("float", 4), ("double", 8)):
    s = bdd_hip_parallel_mma(col, costs, precision=prec)
    s.iterations(30); s.synchronize()
    iters = 300
    ms = s.time_iterations(iters)
    rate = iters / ms * 1e3
    Lp, Vv = s.nr_layers(), s.nr_variables()
    b_iter = 2 * (12 * (N - 2 * Bn) + 2 * R * N + (5 * R + 4) * Lp + (8 * R + 4) * Vv)
    print(f"{name:46s} {prec:6s} BDDs {Bn:8d} nodes {N:9d} hops {s.nr_hops():5d} packs {s.nr_packs():6d} sweeps {s.solve_sweep_kind():10s} "
          f"{rate:8.0f} it/s  {1e3 * ms / iters:8.1f} us/it  frac_whole_iteration {b_iter * rate / 8e12:.3f}  (instance built in {build:.0f} s)", flush=True)
    s.close()
