"""Wide / mixed path benchmark: knapsack rows (general <= rows with non-unit coefficients; layers up to ~77 nodes wide), optionally mixed
with set-cover rows.  Prints sizes, pack statistics, kernel times, iterations/s and the SURVEY §8(d) roofline fraction of the whole
iteration (B_iter = 2 [12 N' + 2R N + (5R+4) L' + (8R+4) V]).

    python tools/widebench.py [--rows 4000] [--vars 20000] [--k 14] [--cover-rows 0] [--cover-k 10] [--precision float,double]
                              [--pack-width 0] [--variant 0] [--wpb 0] [--vars-per-bin 0] [--iters 200]
    4 000 rows = 1 M nodes (the round-1/2 benchmark), 40 000 rows = 10 M nodes."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bdd_amd import capi
if os.environ.get("BDDMMA_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"])
from bdd_amd import native
from bdd_amd.solver import bdd_hip_parallel_mma

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4000)
ap.add_argument("--vars", type=int, default=0, help="default: 5 variables per knapsack row")
ap.add_argument("--k", type=int, default=14)
ap.add_argument("--cover-rows", type=int, default=0)
ap.add_argument("--cover-k", type=int, default=10)
ap.add_argument("--precision", default="float,double")
ap.add_argument("--pack-width", type=int, default=0)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--wide-pack-width", type=int, default=0, dest="wpw")
ap.add_argument("--res", type=int, default=0, help="resident sweeps: 0 auto, 1 off, 2 on")
ap.add_argument("--wpb", type=int, default=0)
ap.add_argument("--vars-per-bin", type=int, default=0)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--stagger", type=int, default=0, help="bddmma_options.pack_stagger: 0 auto, 1 off, N: most hops of a narrow pack")
ap.add_argument("--keep-order", type=int, default=0, dest="keep", help="bddmma_options.keep_bdd_order (2: the pack order before round 5)")
a = ap.parse_args()
V = a.vars or 5 * a.rows
rng = np.random.Generator(np.random.PCG64(1))
t0 = time.time()
rows = []
for _ in range(a.rows):
    vs = np.sort(rng.choice(V, size=a.k, replace=False))
    co = rng.integers(1, 30, size=a.k)
    rows.append((co, vs, "<=", int(co.sum() // 2)))
for _ in range(a.cover_rows):
    rows.append((np.ones(a.cover_k, int), np.sort(rng.choice(V, size=a.cover_k, replace=False)), ">=", 1))
col = native.rows_to_bdd_collection(rows)
ins = col.instr
N = col.nr_bdd_nodes()
Nt = int((ins[:, 2] < 2**62).sum())
print(f"built {col.nr_bdds()} BDDs ({a.rows} knapsack rows of {a.k} variables + {a.cover_rows} covering rows of {a.cover_k}), {N} nodes in {time.time() - t0:.1f} s")
costs = -rng.uniform(1, 10, col.nr_variables())
# pack statistics from the host-side layout (no GPU needed): lane utilisation = node slots in use / (pack,hop) records x pack width
import ctypes as C
Lh = capi.lib()
h = C.c_void_p()
opts = capi.Options(a.pack_width, a.wpw, 0, a.vars_per_bin, 0, a.wpb)
opts.pack_stagger = a.stagger
opts.keep_bdd_order = a.keep
capi.check(Lh.bddmma_layout_create(C.byref(h), np.ascontiguousarray(col.instr).ctypes.data_as(C.c_void_p), np.ascontiguousarray(col.delims).ctypes.data_as(C.c_void_p),
                                   col.nr_bdds(), C.byref(opts)), None)
sz = lambda w: int(Lh.bddmma_layout_size(h, w))
print(f"layout: {sz(3)} narrow packs of width {sz(16)} over {sz(7)} (pack, hop) records holding {sz(1)} slots -> lane utilisation {sz(1) / max(1, sz(7) * sz(16)):.2f}; "
      f"{sz(4)} wide packs over {sz(8)} records holding {sz(0) - sz(1)} slots; {sz(5)} hops")
Lh.bddmma_layout_destroy(h)
for prec in a.precision.split(","):
    R = 4 if prec == "float" else 8
    s = bdd_hip_parallel_mma(col, costs, precision=prec, pack_width=a.pack_width, wide_pack_width=a.wpw, variant_flags=a.variant, resident_sweeps=a.res, waves_per_block=a.wpb, vars_per_bin=a.vars_per_bin, pack_stagger=a.stagger, keep_bdd_order=a.keep)
    L, Vs = s.nr_layers(), s.nr_variables()
    b_iter = 2 * (12 * Nt + 2 * R * N + (5 * R + 4) * L + (8 * R + 4) * Vs)
    s.iterations(5)
    line = f"{prec}: packs {s.nr_packs()}, hops {s.nr_hops()}, layers {L}, variables {Vs}, lb {s.lower_bound():.6f}"
    print(line)
    names = ["fwd_plain", "bwd_plain", "fwd_solve", "bwd_solve", "exchange"]
    print("   " + "  ".join(f"{n}={s.time_kernel(i, 20) * 1e3:.1f}us" for i, n in enumerate(names)))
    ms = s.time_iterations(a.iters) / a.iters
    print(f"   iteration {ms * 1e3:.1f} us = {1e3 / ms:.0f} it/s;  B_iter {b_iter / 1e6:.1f} MB -> {b_iter / ms / 1e9:.3f} TB/s = {b_iter / ms / 1e9 / 8.0:.3f} of 8 TB/s")
