"""of the entries one sweep workgroup stages consecutively (a run of the entry array = one (bin, workgroup) pair), how
many share their variable with another entry of the same run?  CPU only (host layout):  python tools/run_sharing.py [--vars V --rows B]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from bdd_amd.instances import random_set_cover_mt
from test_layout import Layout as Lay
ap = argparse.ArgumentParser()
ap.add_argument("--vars", type=int, default=1_000_000)
ap.add_argument("--rows", type=int, default=500_000)
a = ap.parse_args()
col, _ = random_set_cover_mt(a.vars, a.rows, 10, 12345)
L = Lay(col)
e = L.cs_entry.astype(np.int64)               # staged items, ascending inside a quad's round
ptr = L.cs_ptr.astype(np.int64)
var = L.evar.astype(np.int64)                 # variable of every entry
brk = np.ones(len(e), bool)
brk[1:] = e[1:] != e[:-1] + 1                 # a run ends where the next staged entry is not the next entry
brk[ptr[:-1][ptr[:-1] < len(e)]] = True       # ... or where the next round starts
run_id = np.cumsum(brk) - 1
n_runs = run_id[-1] + 1
lens = np.bincount(run_id)
key = run_id * (a.vars + 1) + var[e]
u, cnt = np.unique(key, return_counts=True)
shared = int(cnt[cnt > 1].sum())
print(f"V = {a.vars}, {a.rows} rows: {len(e)} staged entries in {n_runs} runs (mean {lens.mean():.2f}, median {np.median(lens):.0f}, max {lens.max()}); "
      f"entries that share their variable with another entry of their run: {shared} = {100.0 * shared / len(e):.2f} %; "
      f"pairs a (run, variable) broadcast would save: {len(e) - len(u)} = {100.0 * (len(e) - len(u)) / len(e):.2f} %")
