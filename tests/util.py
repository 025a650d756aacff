"""Shared helpers of the test-suite."""
import glob
import os

import numpy as np

from bdd_amd.bdd_collection import BddCollection

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_ALL = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
GOLDEN = [n for n in _ALL if not n.startswith(("split_", "fullsize_"))]   # MMA traces (oracle/make_golden.py: record)
FULLSIZE = os.path.join(GOLDEN_DIR, "fullsize_set_cover_mt.npz")  # lower-bound trajectories of the reference-compiled code at BASELINE.json's full sizes
SPLIT_GOLDEN = [n for n in _ALL if n.startswith("split_")]      # split_qbdd input/output pairs (record_split)


def collection_from_arrays(instr, delims):
    return BddCollection.from_arrays(instr, delims)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    return BddCollection.from_arrays(z["instr"], z["delims"]), z


def pad_costs(costs, n):
    c = np.zeros(n)
    c[: len(costs)] = costs
    return c


def suffix(precision):
    return "f64" if precision == "double" else "f32"


def same_function(col_a, b_a, col_b, b_b, n_vars, max_enum=14):
    """Do two BDDs represent the same Boolean function over their variables?"""
    va = col_a.variables(b_a)
    assert va == col_b.variables(b_b)
    k = len(va)
    assert k <= max_enum
    x = [0] * n_vars
    for m in range(1 << k):
        for i, v in enumerate(va):
            x[v] = (m >> i) & 1
        if col_a.evaluate(b_a, x) != col_b.evaluate(b_b, x):
            return False
    return True


def canonical_nodes(col, b):
    """BDD b with its nodes renumbered by breadth-first discovery from the root (lo before hi): a list of
    (variable, lo, hi) with 'T' / 'B' for the sinks.  Two BDDs are isomorphic iff these lists are equal — the order of
    the nodes inside a layer carries no meaning (and differs between the reference's bdd_mgr-based constructions and
    the closed-form ones here)."""
    ins = col.instr
    root = int(col.delims[b])
    ids, order = {root: 0}, [root]
    out = []
    k = 0
    while k < len(order):
        i = order[k]
        k += 1
        row = []
        for c in (int(ins[i, 0]), int(ins[i, 1])):
            t = int(ins[c, 2])
            if t == 2**64 - 1:
                row.append("T")
            elif t == 2**64 - 2:
                row.append("B")
            else:
                if c not in ids:
                    ids[c] = len(order)
                    order.append(c)
                row.append(ids[c])
        out.append((int(ins[i, 2]), row[0], row[1]))
    return out
