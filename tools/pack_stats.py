"""Pack statistics of tools/widebench.py's instances from the host-side layout alone (no GPU): packs, (pack, hop) records, lane utilisation,
for keep_bdd_order = 0 (diamond-shaped BDDs widest first), 2 (grouped by shape only: the order before round 5) and 1 (input order).

    python tools/pack_stats.py [--rows 4000] [--k 14] [--cover-rows 0] [--cover-k 10] [--stagger 0] [--pack-width 0]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from bdd_amd import capi, native
ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=4000)
ap.add_argument("--k", type=int, default=14)
ap.add_argument("--cover-rows", type=int, default=0)
ap.add_argument("--cover-k", type=int, default=10)
ap.add_argument("--stagger", type=int, default=0)
ap.add_argument("--pack-width", type=int, default=0)
a = ap.parse_args()
V = 5 * a.rows
rng = np.random.Generator(np.random.PCG64(1))
rows = []
for _ in range(a.rows):
    vs = np.sort(rng.choice(V, size=a.k, replace=False))
    co = rng.integers(1, 30, size=a.k)
    rows.append((co, vs, "<=", int(co.sum() // 2)))
for _ in range(a.cover_rows):
    rows.append((np.ones(a.cover_k, int), np.sort(rng.choice(V, size=a.cover_k, replace=False)), ">=", 1))
col = native.rows_to_bdd_collection(rows)
print(f"{col.nr_bdds()} BDDs, {col.nr_bdd_nodes()} nodes")
Lh = capi.lib()
for keep in (0, 2, 1):
    h = C.c_void_p()
    opts = capi.Options(a.pack_width, 0, 0, 0, 0, 0)
    opts.pack_stagger = a.stagger
    opts.keep_bdd_order = keep
    capi.check(Lh.bddmma_layout_create(C.byref(h), np.ascontiguousarray(col.instr).ctypes.data_as(C.c_void_p), np.ascontiguousarray(col.delims).ctypes.data_as(C.c_void_p),
                                       col.nr_bdds(), C.byref(opts)), None)
    sz = lambda w: int(Lh.bddmma_layout_size(h, w))
    print(f"keep_bdd_order={keep}: {sz(3)} narrow packs of width {sz(16)} over {sz(7)} (pack, hop) records holding {sz(1)} slots -> lane utilisation "
          f"{sz(1) / max(1, sz(7) * sz(16)):.3f}; {sz(4)} wide packs over {sz(8)} records holding {sz(0) - sz(1)} slots; {sz(5)} hops")
    Lh.bddmma_layout_destroy(h)
