"""The driver's text exports ("export bdd lp", "export bdd graph"; reference src/bdd_solver/bdd_solver.cpp:400-410, :432-462) against the
reference's own output on the same collection (tests/golden/exports.json, written by oracle/make_golden.py --exports from the reference's
bdd_collection::write_bdd_lp / ::export_graphviz compiled where they lie): the C++ host code through the C-ABI byte for byte, the Python twin
byte for byte for the LP and up to the (meaningless, hash-ordered) order of the clusters for the graphs.  CPU only."""
import json
import os

import numpy as np

from bdd_amd import BddCollection, native
from util import GOLDEN_DIR


def load():
    z = json.load(open(os.path.join(GOLDEN_DIR, "exports.json")))
    col = BddCollection.from_arrays(np.array(z["instr"], dtype=np.uint64), np.array(z["delims"], dtype=np.uint64))
    return z, col


def test_write_bdd_lp_matches_the_reference(tmp_path):
    z, col = load()
    path = str(tmp_path / "bdd.lp")
    native.write_bdd_lp(col, z["costs"], path)
    assert open(path).read() == z["bdd_lp"]
    assert col.write_bdd_lp(z["costs"]) == z["bdd_lp"]
    # every arc that can carry flow is a binary, every BDD has its root row, every (BDD, variable) pair its linking row
    lp = z["bdd_lp"]
    assert lp.count("\nR_") == col.nr_bdds()
    assert sum(1 for l in lp.splitlines() if l.startswith(" + arc") and " - x_" in l) == sum(len(col.variables(b)) for b in range(col.nr_bdds()))


def test_export_graphviz_matches_the_reference(tmp_path):
    z, col = load()
    for b, ref in enumerate(z["graphviz"]):
        path = str(tmp_path / f"g_{b}.dot")
        native.export_graphviz(col, b, path)
        assert open(path).read() == ref                      # same containers, same insertion order: the same file
        twin = col.export_graphviz(b)
        assert sorted(twin.splitlines()) == sorted(ref.splitlines())   # the twin orders the clusters by variable
        assert twin.startswith("digraph BDD\n{\n") and twin.endswith("}\n")


def test_driver_writes_the_exports(tmp_path):
    """the Python driver up to the point where it needs a device: the export files are there, with the collection's content"""
    from bdd_amd.bdd_solver import bdd_solver
    lp = "Minimize\n1 x_1 + 2 x_2 + 1.5 x_3\nSubject To\nx_1 + x_2 + x_3 = 1\nx_2 + x_3 >= 1\nEnd\n"
    src = tmp_path / "p.lp"
    src.write_text(lp)
    cfg = {"input": str(src), "relaxation solver": "cuda parallel mma", "export bdd lp": str(tmp_path / "out.lp"),
           "export bdd graph": str(tmp_path / "graph.dot")}
    try:
        bdd_solver(cfg).solve()
    except Exception:
        pass    # no HIP device on the CPU box: the solver construction fails after the exports were written
    out = (tmp_path / "out.lp").read_text()
    assert out.startswith("Minimize\n+1 x_0\n+2 x_1\n+1.5 x_2\nSubject To\nR_0: ") and out.endswith("End\n")
    assert (tmp_path / "graph_0.dot").read_text().startswith("digraph BDD\n{\n") and (tmp_path / "graph_1.dot").exists()


def test_exports_refuse_malformed_collections(tmp_path):
    """the emitters index the arrays directly, so the entry points validate what bddmma_create validates — a malformed
    collection is an error return, never an out-of-bounds access"""
    import pytest
    z, col = load()
    good_i, good_d = col.instr.copy(), col.delims.copy()
    path = str(tmp_path / "x.out")

    def broken(instr=None, delims=None):
        c = BddCollection.from_arrays(good_i, good_d)      # (from_arrays validates as well: the damage goes in behind it)
        if instr is not None:
            c._chunks = [instr]
        if delims is not None:
            c._delims = [delims]
        return c

    cases = []
    i = good_i.copy(); i[0, 0] = len(good_i) + 5; cases.append(broken(instr=i))                    # lo child outside the array
    i = good_i.copy(); i[0, 1] = 0; cases.append(broken(instr=i))                                   # hi child points backwards (at itself)
    i = good_i.copy(); i[int(good_d[1]) - 1, 2] = i[int(good_d[1]) - 2, 2]; cases.append(broken(instr=i))   # two equal sinks
    d = good_d.copy(); d[1] = d[0] + 2; cases.append(broken(delims=d))                               # a BDD of two instructions
    d = good_d.copy(); d[1], d[2] = d[2], d[1]; cases.append(broken(delims=d))                       # delimiters not ascending
    for bad in cases:
        with pytest.raises((RuntimeError, ValueError)):
            native.write_bdd_lp(bad, z["costs"], path)
        with pytest.raises((RuntimeError, ValueError)):
            native.export_graphviz(bad, 0, path)
