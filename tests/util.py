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


# ---- checkpoint files (capi.cpp: bddmma_save / bddmma_load) --------------------------------------------------------------------
# layout.hpp: visit_layout_arrays ids
CHECKPOINT_ARRAY_IDS = dict(narrow_words=1, layer_var=4, num_bdds_per_var=6, var_layers=8, bdd_root_slot=9, evar=22, bvar=23, lpos=24, vpos=25,
                            bin_ptr=26, grp_layer_off=28, grp_hop_end=29, cs_entry=32, cs_slot=33, pack_hdr=34, quad_hdr=35, nodes_per_hop=36,
                            layers_per_hop=37)


class Checksum:
    """capi.cpp: struct Checksum (four multiply-rotate lanes over 8-byte words), so that the test can write files whose checksum is
    right and whose contents are not"""
    M = (1 << 64) - 1

    def __init__(self):
        self.lane = [0x9E3779B97F4A7C15, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9, 0x27D4EB2F165667C5]
        self.n = 0

    @classmethod
    def mix(cls, h, w):
        h ^= (w * 0x9FB21C651E98DF25) & cls.M
        h = ((h << 27) | (h >> 37)) & cls.M
        return (h * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & cls.M

    def add(self, b):
        self.n += len(b)
        full = len(b) // 32 * 32
        words = np.frombuffer(b[:full], dtype="<u8").tolist()
        for i in range(0, len(words), 4):
            for k in range(4):
                self.lane[k] = self.mix(self.lane[k], words[i + k])
        k = 0
        for i in range(full, len(b), 8):
            w = int.from_bytes(b[i:i + 8].ljust(8, b"\0"), "little") ^ 0xA5A5A5A5A5A5A5A5
            self.lane[k & 3] = self.mix(self.lane[k & 3], w)
            k += 1

    def value(self):
        h = self.n
        for l in self.lane:
            h = self.mix(h, l)
        return h


def parse_checkpoint(raw):
    """-> (prefix bytes up to the scalars, scalars bytes, options bytes, [(id, esize, count, data)], stored checksum, tail)"""
    n_arrays = int.from_bytes(raw[16:24], "little")
    sc_size, opt_size = int.from_bytes(raw[24:32], "little"), int.from_bytes(raw[32:40], "little")
    pos = 40
    sc = raw[pos:pos + sc_size]; pos += sc_size
    opts = raw[pos:pos + opt_size]; pos += opt_size
    recs = []
    for _ in range(n_arrays):
        i, es, cnt = (int.from_bytes(raw[pos + 8 * j:pos + 8 * j + 8], "little") for j in range(3))
        pos += 24
        recs.append([i, es, cnt, raw[pos:pos + es * cnt]])
        pos += es * cnt
    return raw[:40], sc, opts, recs, int.from_bytes(raw[pos:pos + 8], "little"), raw[pos + 8:]


def write_checkpoint(path, head, sc, opts, recs, tail):
    cs = Checksum()
    cs.add(sc)
    cs.add(opts)
    body = b""
    for i, es, cnt, data in recs:
        rec = i.to_bytes(8, "little") + es.to_bytes(8, "little") + cnt.to_bytes(8, "little")
        cs.add(rec); cs.add(data)
        body += rec + data
    open(path, "wb").write(head + sc + opts + body + cs.value().to_bytes(8, "little") + tail)


