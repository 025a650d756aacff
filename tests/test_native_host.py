"""The C++ host-side input stage (include/bdd_ilp.h: .lp reader, row -> QBDD converter, splitting) against the
Python specification of the same stage, which tests/test_bdd_builders.py, test_lp_reader.py and test_split_qbdd.py
pin against the reference — CPU only."""
import numpy as np
import pytest

from bdd_amd import native
from bdd_amd.bdd_collection import BddCollection
from bdd_amd.ilp import parse_lp, split_long_bdds, to_bdd_collection
from bdd_amd.instances import GRID_3X3, LONG_CHAIN, SHORT_CHAIN, assignment_ilp, mrf_ilp

LPS = {
    "covering": "Minimize\nx1 + x2 + x3 + x4 + x5 + x6\nSubject To\nx1 + x2 + x4 >= 1\nx1 + x3 + x5 >= 1\nx2 + x3 + x6 >= 1\n"
                "Bounds\nBinaries\nx1\nx2\nx3\nx4\nx5\nx6\nEnd\n",
    "names_and_signs": "\\ a comment\nMinimize\nobj: 2 a - 3.5 b + c_1 - 1e1 d(2) + 4\nSubject To\nr1: a + b + c_1 = 1\n"
                       "r2: - a + 2 b\n - d(2) <= 1\n3 a - 2 c_1 >= -1\nEnd\n",
    "multiline": "minimize\n x + y\n + z\nsubject to\n c: x\n + y\n + z >= 2\n x - y = 0\nend\n",
}


def same_ilp(a, b):
    assert a.var_names == b.var_names
    np.testing.assert_allclose(a.objective, b.objective, rtol=0, atol=0)
    assert a.constant == b.constant
    assert len(a.constraints) == len(b.constraints)
    for ca, cb in zip(a.constraints, b.constraints):
        assert (ca.coefficients, ca.variables, ca.ineq, ca.rhs, ca.name) == (cb.coefficients, cb.variables, cb.ineq, cb.rhs, cb.name)


@pytest.mark.parametrize("name", sorted(LPS))
def test_lp_reader_matches_python(name):
    same_ilp(native.parse_lp(LPS[name]), parse_lp(LPS[name]))


@pytest.mark.parametrize("make", [lambda: assignment_ilp(4), lambda: mrf_ilp(**SHORT_CHAIN), lambda: mrf_ilp(**LONG_CHAIN),
                                  lambda: mrf_ilp(**GRID_3X3)])
def test_written_lp_round_trips_through_both_readers_and_converters(make):
    ilp = make()
    text = ilp.write_lp()
    same_ilp(native.parse_lp(text), parse_lp(text))
    want = to_bdd_collection(parse_lp(text))
    got = native.lp_to_bdd_collection(text)
    np.testing.assert_array_equal(got.delims, want.delims)
    np.testing.assert_array_equal(got.instr, want.instr)


@pytest.mark.parametrize("bad,match", [("Subject To\nx >= 1\nEnd\n", "Minimize"), ("Minimize\nx\nEnd\n", "Subject To"),
                                       ("Minimize\nx\nSubject To\nx + 0.5 y >= 1\nEnd\n", "integer"),
                                       ("Minimize\nx\nSubject To\nx + y >=\nEnd\n", "incomplete")])
def test_reader_errors(bad, match):
    with pytest.raises(ValueError, match=match):
        native.parse_lp(bad)
    with pytest.raises(ValueError, match=match):
        parse_lp(bad)


def test_random_rows_node_for_node():
    rng = np.random.Generator(np.random.PCG64(17))
    rows, want = [], BddCollection()
    while len(rows) < 150:
        k = int(rng.integers(1, 11))
        vs = rng.choice(40, size=k, replace=False)           # unsorted on purpose: the row's own order is the BDD order
        co = rng.integers(-5, 6, size=k)
        co[co == 0] = 1
        rhs = int(rng.integers(co[co < 0].sum() - 1, co[co > 0].sum() + 2))
        ineq = ["<=", "=", ">="][int(rng.integers(0, 3))]
        try:
            if ineq == "=" and rhs != 0 and all(c == rhs for c in co):
                want.add_simplex(vs)
            else:
                want.add_linear(co, ineq, rhs, vs)
        except ValueError as e:
            if "infeasible" in str(e):
                with pytest.raises(RuntimeError, match="infeasible"):
                    native.rows_to_bdd_collection([(co, vs, ineq, rhs)])
                continue
            assert "trivially true" in str(e)                  # skipped by both
        rows.append((co, vs, ineq, rhs))
    got = native.rows_to_bdd_collection(rows)
    assert got.nr_bdds() == want.nr_bdds() > 50
    np.testing.assert_array_equal(got.delims, want.delims)
    np.testing.assert_array_equal(got.instr, want.instr)


@pytest.mark.parametrize("split_length", [2, 3, 7])
def test_splitting_node_for_node(split_length):
    rng = np.random.Generator(np.random.PCG64(23))
    rows, want = [], BddCollection()
    for _ in range(12):
        k = int(rng.integers(5, 16))
        vs = np.sort(rng.choice(30, size=k, replace=False))
        co = rng.integers(1, 5, size=k)
        rows.append((co, vs, "<=", int(co.sum() // 2)))
        want.add_linear(co, "<=", int(co.sum() // 2), vs)
    # avoid width-1 cut layers? no: both implementations share the width-1 extension
    split_long_bdds(want, 30, split_length)
    got = native.rows_to_bdd_collection(rows, split_length=split_length, nr_variables=30)
    np.testing.assert_array_equal(got.delims, want.delims)
    np.testing.assert_array_equal(got.instr, want.instr)


def test_splitting_with_implication_bdd_node_for_node():
    rows, want = [], BddCollection()
    for co, rhs in (([1, 2, 3, 2, 1, 3, 2, 1, 2, 3, 1, 2], 11), ([3, 1, 4, 1, 5, 2, 6, 5, 3, 5, 2, 3], 18), ([1] * 12, 5)):
        rows.append((co, np.arange(12), "<=", rhs))
        want.add_linear(co, "<=", rhs, np.arange(12))
    split_long_bdds(want, 12, 3, with_implication_bdd=True)
    got = native.rows_to_bdd_collection(rows, split_length=3, nr_variables=12, with_implication_bdd=True)
    assert got.nr_bdds() == 3 * 4 + 3
    np.testing.assert_array_equal(got.delims, want.delims)
    np.testing.assert_array_equal(got.instr, want.instr)


def test_infeasible_and_trivial_rows_in_lp():
    with pytest.raises(RuntimeError, match="infeasible"):
        native.lp_to_bdd_collection("Minimize\nx + y\nSubject To\nx + y >= 3\nEnd\n")
    col = native.lp_to_bdd_collection("Minimize\nx + y\nSubject To\nx + y >= 0\nx + y >= 1\nEnd\n")
    assert col.nr_bdds() == 1


OPB = """* #variable= 5 #constraint= 3
* a comment
min: +1 x1 -2 x2 +3.5 x3 + x4 - x5 ;
+1 x1 +2 x2 >= 1 ;
-1 x3 +1 x4
 +1 x5 = 1 ;
3 x1 - x5 <= 2;
"""


def test_opb_reader_cpp_python_and_lp_equivalent():
    from bdd_amd.ilp import parse_lp_or_opb, parse_opb
    a, b = native.parse_lp(OPB, fmt="opb"), parse_opb(OPB)
    same_ilp(a, b)
    assert a.var_names == ["x1", "x2", "x3", "x4", "x5"] and a.objective == [1.0, -2.0, 3.5, 1.0, -1.0]
    rows = [(c.coefficients, c.variables, c.ineq, c.rhs) for c in a.constraints]
    assert rows == [([1, 2], [0, 1], ">=", 1), ([-1, 1, 1], [2, 3, 4], "=", 1), ([3, -1], [0, 4], "<=", 2)]
    lp = "Minimize\nx1 - 2 x2 + 3.5 x3 + x4 - x5\nSubject To\nx1 + 2 x2 >= 1\n- x3 + x4 + x5 = 1\n3 x1 - x5 <= 2\nEnd\n"
    same_ilp(a, parse_lp(lp))
    # the driver's fallback: .lp grammar first, then OPB; the .lp diagnosis is reported when both fail
    same_ilp(native.parse_lp(OPB, fmt="auto"), a)
    same_ilp(parse_lp_or_opb(OPB), a)
    same_ilp(native.parse_lp(lp, fmt="auto"), a)
    for f in (lambda t: native.parse_lp(t, fmt="auto"), parse_lp_or_opb):
        with pytest.raises(ValueError, match="Minimize"):
            f("max: x1 ;")
    for f in (lambda t: native.parse_lp(t, fmt="opb"), parse_opb):
        with pytest.raises(ValueError, match="min:"):
            f("x1 + x2 >= 1 ;")
        with pytest.raises(ValueError, match="integer"):
            f("min: x1 ;\n x1 + 0.5 x2 >= 1 ;")


def test_conversion_is_linear_in_the_number_of_rows():
    """100 000 small rows convert in seconds (a reserve() per appended BDD once made this quadratic: 15 minutes for 250 000 rows)."""
    import time
    rng = np.random.Generator(np.random.PCG64(1))
    vs = rng.integers(0, 50_000, size=(100_000, 3))
    vs[:, 1] += 50_000; vs[:, 2] += 100_000                      # three distinct variables per row
    rows = [((1, 1, 1), v, "=", 1) for v in vs]
    t = time.perf_counter()
    col = native.rows_to_bdd_collection(rows)
    dt = time.perf_counter() - t
    assert col.nr_bdds() == 100_000 and col.nr_bdd_nodes() == 100_000 * 7
    assert dt < 20.0, dt
