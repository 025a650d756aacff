"""Pin the CPU oracle (oracle/mma_oracle.c) — CPU only.

(a) known-answer values of the reference's own tests for this path,
(b) golden traces produced by the reference's compiled node code (oracle/_ref via oracle/make_golden.py),
(c) the lower-bound trajectory recorded from the unmodified reference in BASELINE.md §2.
"""
import numpy as np
import pytest

from bdd_amd import parse_lp, to_bdd_collection
from bdd_amd.instances import GRID_3X3, LONG_CHAIN, SHORT_CHAIN, assignment_ilp, brute_force_optimum, mrf_ilp
from oracle.oracle import Oracle
from util import GOLDEN, load_golden, pad_costs, suffix


def solve(ilp, iters, precision="double"):
    col = to_bdd_collection(ilp)
    o = Oracle(col, pad_costs(ilp.objective, max(col.nr_variables(), ilp.nr_variables())), precision)
    lb0 = o.lower_bound()
    for _ in range(iters):
        o.iteration()
    return lb0, o.lower_bound(), o


def test_matching_3x3_diag():
    # test/test_bdd_cuda_base.cpp:100-114 (lb == -6), test_bdd_bipartite_matching_problem.cpp:57
    lb0, lb, _ = solve(assignment_ilp(3), 20)
    assert lb0 == -6.0 and lb == -6.0


def test_matching_3x3_first_row_trajectory():
    # trajectory of the unmodified reference CPU parallel mma (double)
    c = -np.ones((3, 3)); c[:, 0] = -2
    ilp = assignment_ilp(3, c)
    col = to_bdd_collection(ilp)
    o = Oracle(col, ilp.objective)
    assert o.lower_bound() == -5.0
    expect = [-4.78125, -4.3671875, -4.187866210938, -4.103347778320, -4.059029579163]
    for i in range(100):
        o.iteration()
        if i < 5:
            assert abs(o.lower_bound() - expect[i]) < 1e-12  # values printed with 12 decimals
        if i == 49:
            assert abs(o.lower_bound() - (-4.000000000057)) < 1e-12
    assert abs(o.lower_bound() - (-4.0)) < 1e-12  # test_bdd_bipartite_matching_problem.cpp:58


def test_matching_8x8():
    # BASELINE.json config 1: n x n assignment, optimum -2n
    lb0, lb, _ = solve(assignment_ilp(8), 20)
    assert abs(lb - (-16.0)) < 1e-6


SIMPLEX_KATS = [
    # test/test_bdd_cuda_base.cpp:10-34, 49-98
    ("Minimize\n-2 x_11 - 1 x_12 - 1 x_13\nSubject To\nx_11 + x_12 + x_13 = 1\nEnd\n", 3, 1, -2.0),
    ("Minimize\n2 x_1 + 1 x_2 + 1 x_3\n+1 x_4 + 2 x_5 - 1 x_6\nSubject To\nx_1 + x_2 + x_3 = 1\nx_4 + x_5 + x_6 = 2\nEnd\n", 6, 2, 1.0),
    ("Minimize\n2 x_1 + 1 x_2 + 1 x_3\n+2 x_4 + 2 x_5 + 3 x_6\nSubject To\nx_1 + x_2 + x_3 + x_4 = 1\nx_4 + x_5 + x_6 = 2\nEnd\n", 6, 2, 4.0),
]


@pytest.mark.parametrize("lp,nv,nb,lb", SIMPLEX_KATS)
def test_bdd_cuda_base_kats(lp, nv, nb, lb):
    ilp = parse_lp(lp)
    col = to_bdd_collection(ilp)
    for prec in ("float", "double"):
        o = Oracle(col, ilp.objective, prec)
        assert o.nr_variables() == nv and o.nr_bdds() == nb
        assert o.lower_bound() == lb


def test_two_simplex_min_marginals():
    # test/test_bdd_cuda_min_marginals.cpp:17-36
    ilp = parse_lp(SIMPLEX_KATS[1][0])
    o = Oracle(to_bdd_collection(ilp), ilp.objective)
    mm = o.min_marginals()
    var, _ = o.layer_info()
    expect = {0: (1, 2), 1: (1, 1), 2: (1, 1), 3: (1, 0), 4: (0, 1), 5: (3, 0)}
    for l in range(6):
        assert tuple(mm[l]) == expect[int(var[l])]


def test_loose_covering():
    # test/test_loose_covering_problem.cpp:8-22,59 (LB 1.5)
    lp = ("Minimize\nx1 + x2 + x3 + x4 + x5 + x6\nSubject To\nx1 + x2 + x4 >= 1\nx1 + x3 + x5 >= 1\n"
          "x2 + x3 + x6 >= 1\nBounds\nBinaries\nx1\nx2\nx3\nx4\nx5\nx6\nEnd\n")
    lb0, lb, _ = solve(parse_lp(lp), 200)
    assert abs(lb - 1.5) <= 1e-4 and abs(lb0 - 1.5) <= 1e-12


@pytest.mark.parametrize("P,kat", [(SHORT_CHAIN, 1.0), (LONG_CHAIN, -9.0), (GRID_3X3, -8.0)])
def test_mrf_kats(P, kat):
    # test/test_bdd_cuda_parallel_mma.cu:197-247: 200 iterations, final LB 1 / -9 / -8
    ilp = mrf_ilp(**P)
    lb0, lb, o = solve(ilp, 200)
    assert abs(lb - kat) < 1e-9
    assert lb0 <= lb + 1e-12
    if ilp.nr_variables() <= 22:
        assert abs(brute_force_optimum(ilp) - kat) < 1e-9  # LP relaxation of a tree MRF is tight


def test_mma_monotone_and_omega_one_single_bdd():
    # test/test_bdd_parallel_mma.cpp:19-60: on a single BDD a second sweep produces zero differences
    from bdd_amd import BddCollection
    col = BddCollection()
    col.add_simplex([0, 1, 2, 3])
    o = Oracle(col, [3.0, -1.0, 2.0, 0.5])
    lb = o.lower_bound()
    assert lb == -1.0
    for _ in range(3):
        o.iteration()
        assert abs(o.lower_bound() - lb) < 1e-12


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("precision", ["double", "float"])
def test_oracle_matches_reference_traces(name, precision):
    """Bit-for-bit agreement with traces produced by the reference's compiled node arithmetic."""
    col, z = load_golden(name)
    sfx = suffix(precision)
    dt = np.float64 if precision == "double" else np.float32
    o = Oracle(col, None, precision)
    V = o.nr_variables()
    o.update_costs([], pad_costs(z["costs"], V))
    assert o.lower_bound() == float(z[f"lb_init_{sfx}"])
    d = np.zeros(2 * V, dt)
    for it in range(10):
        o.forward_mm(0.5, d)
        np.testing.assert_array_equal(d, z[f"delta_trace_{sfx}"][it, 0])
        lb = o.backward_mm(0.5, d)
        np.testing.assert_array_equal(d, z[f"delta_trace_{sfx}"][it, 1])
        assert lb == float(z[f"lb_trace_{sfx}"][it])
        assert o.lower_bound() == lb
    o2 = Oracle(col, None, precision)
    o2.update_costs([], pad_costs(z["costs"], V))
    for it in range(20):
        o2.iteration()
        assert o2.lower_bound() == float(z[f"iter_lb_{sfx}"][it])


def test_oracle_threads_agree():
    col, z = load_golden("random_cover_small")
    a = Oracle(col, None, "double", threads=1)
    b = Oracle(col, None, "double", threads=4)
    for o in (a, b):
        o.update_costs([], pad_costs(z["costs"], o.nr_variables()))
    for _ in range(10):
        a.iteration(); b.iteration()
    assert abs(a.lower_bound() - b.lower_bound()) < 1e-9


def test_bdds_solution_and_dual_ops():
    # test/test_bdd_cuda_base_sol.cpp:30-86 analogue on the oracle: per-BDD argmin is feasible and optimal
    ilp = parse_lp(SIMPLEX_KATS[1][0])
    col = to_bdd_collection(ilp)
    o = Oracle(col, ilp.objective)
    sol = o.bdds_solution_vec()
    var, bdd = o.layer_info()
    x = np.zeros(6)
    x[var] = sol
    assert col.evaluate(0, x) and col.evaluate(1, x)
    assert abs(np.dot(x, ilp.objective) - o.lower_bound()) < 1e-12
    g = np.arange(o.nr_layers(), dtype=np.float64)
    o.make_dual_feasible(g)
    nb = o.nr_bdds_per_var()
    s = np.zeros(6)
    np.add.at(s, var, g)
    assert np.allclose(s[nb > 0], 0)


@pytest.mark.parametrize("precision", ["double", "float"])
def test_oracle_full_size_vs_reference_compiled_trajectory(precision):
    """BASELINE.json configs[1] (1.05 M nodes): the restated oracle reproduces the lower-bound trajectory that the
    reference-compiled code (oracle/_ref: reference bdd_collection + bdd_branch_instruction arithmetic) produced on the
    mt19937_64(12345) benchmark instance — tests/golden/fullsize_set_cover_mt.npz, oracle/make_golden.py --fullsize."""
    from bdd_amd.instances import random_set_cover_mt
    from util import FULLSIZE
    z = np.load(FULLSIZE)
    n_vars, n_rows, k, seed, iters = (int(x) for x in z["1m_params"])
    col, costs = random_set_cover_mt(n_vars, n_rows, k, seed)
    ref = z["1m_lb_f64" if precision == "double" else "1m_lb_f32"]
    o = Oracle(col, costs, precision, threads=4)
    rel = 1e-11 if precision == "double" else 1e-6   # only the order of the delta sums (and of the bound's summation) differs
    assert abs(o.lower_bound() - ref[0]) <= rel * abs(ref[0])
    for it in range(iters):
        o.iteration()
        assert abs(o.lower_bound() - ref[it + 1]) <= rel * abs(ref[it + 1]), (it, o.lower_bound(), ref[it + 1])
