"""Closed-form QBDD builders and the linear-row converter vs the reference's own bdd_collection — CPU only."""
import numpy as np
import pytest

from bdd_amd import BddCollection
from oracle import oracle as O
from util import same_function

needs_ref = pytest.mark.skipif(not O.ref_available(), reason="oracle/_ref not built (needs /root/reference)")


@needs_ref
@pytest.mark.parametrize("n", [1, 2, 3, 4, 7, 10])
def test_simplex_and_covering_node_for_node(n):
    vs = np.arange(3, 3 + 2 * n, 2)
    mine = BddCollection(); mine.add_simplex(vs); mine.add_covering(vs)
    rc = O.RefCollection(); rc.add_simplex(vs); rc.add_covering(vs)
    ref = rc.export()
    np.testing.assert_array_equal(mine.delims, ref.delims)
    np.testing.assert_array_equal(mine.instr, ref.instr)


@needs_ref
def test_linear_rows_same_function_and_size():
    rng = np.random.Generator(np.random.PCG64(3))
    for _ in range(40):
        k = int(rng.integers(2, 9))
        vs = np.sort(rng.choice(12, size=k, replace=False))
        co = rng.integers(-4, 5, size=k)
        co[co == 0] = 1
        rhs = int(rng.integers(co[co < 0].sum(), co[co > 0].sum() + 1))
        ineq = ["<=", "=", ">="][int(rng.integers(0, 3))]
        rc = O.RefCollection()
        r = rc.add_linear(co, ineq, rhs, vs)
        mine = BddCollection()
        try:
            mine.add_linear(co, ineq, rhs, vs)
        except ValueError as e:
            assert (r == -1 and "trivially" in str(e)) or (r == -2 and "infeasible" in str(e))
            continue
        assert r >= 0
        ref = rc.export()
        assert same_function(mine, 0, ref, 0, 12)
        assert mine.nr_bdd_nodes() == ref.nr_bdd_nodes()  # both are the canonical (minimal) QBDD


@needs_ref
def test_cardinality_matches_reference():
    for n, k in [(3, 2), (5, 2), (6, 3)]:
        vs = list(range(n))
        rc = O.RefCollection(); rc.add_cardinality(vs, k)
        mine = BddCollection(); mine.add_cardinality(vs, k)
        ref = rc.export()
        assert same_function(mine, 0, ref, 0, n)
        assert mine.nr_bdd_nodes() == ref.nr_bdd_nodes()


def test_builders_evaluate():
    col = BddCollection()
    col.add_simplex([0, 1, 2]); col.add_covering([0, 1, 2]); col.add_linear([2, 3, -1], "<=", 3, [0, 1, 2])
    for m in range(8):
        x = [(m >> i) & 1 for i in range(3)]
        assert col.evaluate(0, x) == (sum(x) == 1)
        assert col.evaluate(1, x) == (sum(x) >= 1)
        assert col.evaluate(2, x) == (2 * x[0] + 3 * x[1] - x[2] <= 3)


def test_batched_builders_match_single():
    rows = np.array([[0, 3, 5], [1, 2, 9], [4, 6, 7]], dtype=np.uint64)
    a = BddCollection(); a.add_covering(rows)
    b = BddCollection()
    for r in rows:
        b.add_covering(r)
    np.testing.assert_array_equal(a.instr, b.instr)
    np.testing.assert_array_equal(a.delims, b.delims)


@needs_ref
def test_dropin_template_constructor_flattens_the_reference_collection():
    """`bdd_hip_parallel_mma<REAL>(const BDD::bdd_collection&, costs)` with the REAL reference type
    (include/bdd_collection/bdd_collection.h:206 `operator()(bdd_nr, offset)`, nr_bdd_nodes, offset).  oracle/ref_driver.cpp compiles
    bdd_amd/csrc/bdd_hip_parallel_mma.hpp against -I/root/reference/include and runs the constructor's flattening loop
    (static flatten(): no GPU) on collections built by the reference; the arrays bddmma_create would receive must equal the
    reference-side export.  Several BDDs, so the storage offsets of the collection differ from the dense positions."""
    rng = np.random.Generator(np.random.PCG64(11))
    rc = O.RefCollection()
    for _ in range(12):
        k = int(rng.integers(2, 8))
        vs = np.sort(rng.choice(20, size=k, replace=False))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            rc.add_simplex(vs)
        elif kind == 1:
            rc.add_covering(vs)
        else:
            co = rng.integers(1, 5, size=k)
            assert rc.add_linear(co, "<=", int(co.sum() // 2), vs) >= 0
    ref = rc.export()
    instr, delims = rc.flatten_dropin()
    assert len(delims) == rc.nr_bdds() + 1 and rc.nr_bdds() == 12
    np.testing.assert_array_equal(delims, ref.delims)
    np.testing.assert_array_equal(instr, ref.instr)
