"""Generate tests/golden/*.npz from the REFERENCE's own compiled code (oracle/_ref).

Run in the build container (needs /root/reference):  python oracle/make_golden.py
Every fixture holds inputs (flat BDD arrays built by the reference's bdd_collection, costs) and
expected outputs produced by the reference's node arithmetic (bdd_branch_instruction.h) driven by
oracle/ref_driver.cpp:
  * `delta_trace`  : the CPU<->GPU parity protocol of test/test_cuda_parallel_mma.cu:13-103 —
                     10x { forward_mm(0.5, d); backward_mm(0.5, d) } with d NOT normalised in between;
                     shape (10, 2, 2V)
  * `lb_trace`     : lower bound after every backward_mm of that protocol, shape (10,)
  * `iter_lb`      : lower bound after each of 20 iteration() calls on a fresh solver, shape (20,)
  * `lb_init`      : lower bound right after update_costs
Fixtures are data only; no reference source text is stored.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bdd_amd.ilp import ILP  # noqa: E402
from bdd_amd.instances import GRID_3X3, LONG_CHAIN, SHORT_CHAIN, assignment_ilp, mrf_ilp  # noqa: E402
from oracle.oracle import RefCollection, RefMma  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def ref_collection_from_ilp(ilp: ILP) -> RefCollection:
    rc = RefCollection()
    for c in ilp.constraints:
        if c.is_simplex():
            rc.add_simplex(c.variables)
        else:
            r = rc.add_linear(c.coefficients, c.ineq, c.rhs, c.variables)
            assert r >= 0, "trivial/infeasible constraint"
    return rc


def record(name, rc: RefCollection, costs):
    col = rc.export()
    out = dict(instr=col.instr, delims=col.delims, costs=np.asarray(costs, np.float64))
    for prec, dt in (("f64", np.float64), ("f32", np.float32)):
        m = RefMma(rc, "double" if prec == "f64" else "float")
        V = m.nr_variables()
        c = np.zeros(V)
        c[: len(costs)] = costs
        m.update_costs([], c)
        out[f"lb_init_{prec}"] = m.lower_bound()
        d = np.zeros(2 * V, dt)
        trace = np.zeros((10, 2, 2 * V), dt)
        lbs = np.zeros(10)
        for it in range(10):
            m.forward_mm(0.5, d)
            trace[it, 0] = d
            lbs[it] = m.backward_mm(0.5, d)
            trace[it, 1] = d
        out[f"delta_trace_{prec}"] = trace
        out[f"lb_trace_{prec}"] = lbs
        m2 = RefMma(rc, "double" if prec == "f64" else "float")
        m2.update_costs([], c)
        out[f"iter_lb_{prec}"] = np.array([m2.iteration() for _ in range(20)])
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "V", V, "bdds", rc.nr_bdds(), "lb_init", out["lb_init_f64"], "iter_lb[-1]", out["iter_lb_f64"][-1])


def record_split(name, build, chunk, aux0, implication=False):
    """bdd_collection::split_qbdd of the reference on BDD 0 (then the original removed): input and output storage.
    The lo / hi fields of terminal entries carry no information (make_qbdd leaves scratch values there); they are
    normalised to the sink marker so that the fixtures compare with array equality."""
    rc = RefCollection()
    build(rc)
    before = rc.export()
    n, next_aux = rc.split_qbdd(0, chunk, aux0, implication)
    after = rc.export()
    out = after.instr.copy()
    term = out[:, 2] >= np.uint64(2**64 - 2)
    out[term, 0] = out[term, 2]
    out[term, 1] = out[term, 2]
    np.savez_compressed(os.path.join(OUT, name + ".npz"), in_instr=before.instr, in_delims=before.delims,
                        chunk=chunk, aux0=aux0, n_new=n, next_aux=next_aux, implication=implication, out_instr=out,
                        out_delims=after.delims)
    print(name, "chunks", n, "next aux", next_aux, "nodes", before.nr_bdd_nodes(), "->", after.nr_bdd_nodes())


def record_fullsize():
    """BASELINE.json's full-size instances (configs[1], configs[2]) through the reference-compiled code: rows from
    bdd_amd.instances.random_set_cover_mt (std::mt19937_64(12345), draw order in bdd_amd/csrc/host/instances.cpp), BDDs built
    by the reference's not_all_false_constraint -> make_qbdd -> rebase, iteration() over the reference's node arithmetic.
    Only the lower bounds are stored (the instance is regenerated from the seed): lb after update_costs and after every
    iteration, double and float."""
    import time
    from bdd_amd import capi
    out = {}
    for tag, V, B, k, iters in (("1m", 100_000, 50_000, 10, 20), ("10m", 1_000_000, 500_000, 10, 10)):
        rows = np.zeros((B, k), np.uint64)
        costs = np.zeros(V)
        assert capi.lib().bddilp_random_set_cover(V, B, k, 12345, rows.ctypes.data, costs.ctypes.data) == 0
        t0 = time.time()
        rc = RefCollection()
        for r in rows:
            rc.add_covering(r)
        print(tag, "reference collection built in", round(time.time() - t0, 1), "s")
        out[f"{tag}_params"] = np.array([V, B, k, 12345, iters], np.int64)
        out[f"{tag}_rows_checksum"] = np.array([int(rows.sum() % (2**61 - 1))], np.int64)
        out[f"{tag}_costs_sum"] = np.array([costs.sum()])
        for prec in ("f64", "f32"):
            m = RefMma(rc, "double" if prec == "f64" else "float")
            m.update_costs([], costs)
            lbs = [m.lower_bound()]
            for _ in range(iters):
                lbs.append(m.iteration())
            out[f"{tag}_lb_{prec}"] = np.array(lbs)
            print(tag, prec, "lb after", iters, "iterations:", repr(lbs[-1]), "time", round(time.time() - t0, 1), "s")
            del m
        del rc
    np.savez_compressed(os.path.join(OUT, "fullsize_set_cover_mt.npz"), **out)


def record_exports():
    """tests/golden/exports.json: the reference's text exports (bdd_collection::write_bdd_lp, ::export_graphviz) of a small collection with
    every constraint family, next to the collection itself — what "export bdd lp" / "export bdd graph" of the driver have to produce."""
    import json
    rc = RefCollection()
    rc.add_simplex([0, 1, 2])
    rc.add_covering([1, 3])
    rc.add_linear([2, 3, 4], "<=", 5, [0, 2, 3])
    rc.add_cardinality([0, 1, 2, 3, 4], 2)
    rc.add_linear([3, 1, 4, 1, 5, 2, 6], ">=", 9, [1, 2, 4, 5, 6, 7, 8])
    rc.add_all_equal([5, 7, 8])
    col = rc.export()
    costs = [1.5, -2.0, 1.0 / 3.0, 4.0, 1e-7, -12345.678, 0.0, 2.0, -0.25]
    out = {"instr": [[int(x) for x in row] for row in col.instr], "delims": [int(x) for x in col.delims], "costs": costs,
           "bdd_lp": rc.write_bdd_lp(costs), "graphviz": [rc.export_graphviz(b) for b in range(rc.nr_bdds())]}
    json.dump(out, open(os.path.join(OUT, "exports.json"), "w"), indent=0)
    print("exports.json:", rc.nr_bdds(), "BDDs,", len(out["bdd_lp"]), "bytes of LP")


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--fullsize" in sys.argv:
        record_fullsize()
        return
    if "--exports" in sys.argv:
        record_exports()
        return
    record_split("split_covering_10", lambda x: x.add_covering(list(range(10))), 3, 10)
    record_split("split_simplex_9", lambda x: x.add_simplex(list(range(9))), 4, 20)
    record_split("split_knapsack_12", lambda x: x.add_linear([1, 2, 3, 2, 1, 3, 2, 1, 2, 3, 1, 2], "<=", 11, list(range(12))), 4, 12)
    record_split("split_equality_12", lambda x: x.add_linear([1, 2, 3, 2, 1, 3, 2, 1, 2, 3, 1, 2], "=", 9, list(range(0, 24, 2))), 5, 30)
    record_split("split_cardinality_11", lambda x: x.add_cardinality(list(range(11)), 4), 2, 11)
    record_split("split_not_needed", lambda x: x.add_covering(list(range(4))), 4, 4)
    # with the implication BDD over the auxiliary variables (bdd_collection.cpp:801-941)
    record_split("split_imp_knapsack_12", lambda x: x.add_linear([1, 2, 3, 2, 1, 3, 2, 1, 2, 3, 1, 2], "<=", 11, list(range(12))), 3, 12, True)
    record_split("split_imp_equality_12", lambda x: x.add_linear([1, 2, 3, 2, 1, 3, 2, 1, 2, 3, 1, 2], "=", 9, list(range(12))), 3, 12, True)
    record_split("split_imp_cardinality_11", lambda x: x.add_cardinality(list(range(11)), 4), 2, 11, True)
    record_split("split_imp_covering_10", lambda x: x.add_covering(list(range(10))), 3, 10, True)
    record_split("split_imp_knapsack_14", lambda x: x.add_linear([3, 1, 4, 1, 5, 2, 6, 5, 3, 5, 2, 3, 2, 3], ">=", 20, list(range(14))), 4, 20, True)
    record_split("split_imp_two_chunks", lambda x: x.add_covering(list(range(6))), 3, 6, True)    # two chunks: no implication BDD
    ilp = assignment_ilp(3)
    record("matching_3x3_diag", ref_collection_from_ilp(ilp), ilp.objective)
    c = -np.ones((3, 3)); c[:, 0] = -2
    ilp = assignment_ilp(3, c)
    record("matching_3x3_first_row", ref_collection_from_ilp(ilp), ilp.objective)
    ilp = assignment_ilp(8)
    record("matching_8x8", ref_collection_from_ilp(ilp), ilp.objective)
    # 3-row set cover of test/test_loose_covering_problem.cpp:8-22
    rc = RefCollection()
    for row in ([0, 1, 3], [0, 2, 4], [1, 2, 5]):
        rc.add_covering(row)
    record("loose_covering", rc, np.ones(6))
    for nm, P in (("mrf_short_chain", SHORT_CHAIN), ("mrf_long_chain", LONG_CHAIN), ("mrf_grid_3x3", GRID_3X3)):
        ilp = mrf_ilp(**P)
        record(nm, ref_collection_from_ilp(ilp), ilp.objective)
    # random set cover with variable gaps (some variables in no BDD), mixed row sizes
    rng = np.random.Generator(np.random.PCG64(7))
    rc = RefCollection()
    V = 60
    for _ in range(45):
        k = int(rng.integers(2, 9))
        rc.add_covering(np.sort(rng.choice(V - 5, size=k, replace=False)))
    costs = rng.uniform(1, 10, V)
    col = rc.export()
    used = np.zeros(V, bool)
    used[col.instr[col.instr[:, 2] < 2**63, 2].astype(int)] = True
    costs[~used] = 0
    record("random_cover_small", rc, costs[: col.nr_variables()])
    # knapsack-like rows with wider layers + cardinality rows
    rc = RefCollection()
    rng = np.random.Generator(np.random.PCG64(11))
    V = 24
    for _ in range(10):
        k = int(rng.integers(4, 10))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        co = rng.integers(1, 6, size=k)
        rhs = int(co.sum() // 2)
        rc.add_linear(co, "<=" if rng.random() < 0.5 else ">=", rhs, vs)
    for _ in range(4):
        vs = np.sort(rng.choice(V, size=5, replace=False))
        rc.add_cardinality(vs, 2)
    record("knapsack_mixed", rc, rng.normal(0, 3, V).round(2))


if __name__ == "__main__":
    main()
