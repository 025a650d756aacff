"""LP-subset reader (src/ILP/ILP_parser.cpp:24-140 grammar) — CPU only."""
import pytest

from bdd_amd import parse_lp
from bdd_amd.instances import assignment_ilp


def test_variable_order_and_terms():
    ilp = parse_lp("\\ comment\nMinimize\n2 a - b + 0.5 c_1\n- d\nSubject To\nr1: a + b >= 1\n- a + 2 c_1 - d <= 0\nEnd\n")
    assert ilp.var_names == ["a", "b", "c_1", "d"]
    assert ilp.objective == [2.0, -1.0, 0.5, -1.0]
    c0, c1 = ilp.constraints
    assert (c0.name, c0.coefficients, c0.variables, c0.ineq, c0.rhs) == ("r1", [1, 1], [0, 1], ">=", 1)
    assert (c1.coefficients, c1.variables, c1.ineq, c1.rhs) == ([-1, 2, -1], [0, 2, 3], "<=", 0)


def test_constraint_only_variables_and_sections():
    ilp = parse_lp("Minimize\nx1\nSubject To\nx1 + y = 1\nBounds\nBinaries\nx1\ny\nEnd\n")
    assert ilp.var_names == ["x1", "y"] and ilp.objective == [1.0, 0.0]
    assert ilp.constraints[0].is_simplex()


def test_roundtrip_write_parse():
    a = assignment_ilp(4)
    b = parse_lp(a.write_lp())
    assert a.var_names == b.var_names and a.objective == b.objective
    assert [(c.coefficients, c.variables, c.ineq, c.rhs) for c in a.constraints] == \
           [(c.coefficients, c.variables, c.ineq, c.rhs) for c in b.constraints]


def test_errors():
    with pytest.raises(ValueError):
        parse_lp("Subject To\nx = 1\nEnd\n")
    with pytest.raises(ValueError):
        parse_lp("Minimize\nx\nSubject To\n1.5 x >= 1\nEnd\n")


# ---------------------------------------------------------------- the reference's own parser vectors (test/test_ILP_parser.cpp:8-33)
ILP_EXAMPLE = """Minimize
x1 + 2*x2 + 1.5 * x3 - 0.5*x4 - x5
Subject To
x1 + 2*x2 + 3 * x3 - 5*x4 - x5 >= 1
Bounds
 x1 <= 1
 x2 >= 0
End"""

ILP_EXAMPLE_HASH = """Minimize
x1 + 2*x2 + 1.5 * x#3 - 0.5*x#4 - x3
Subject To
 b_cuta_;0;14_1;@41d: x1 + 2*x2 + 3 * x#3 - 5*x#4 - x3 >= 1
Bounds
 x1 <= 1
 x2 >= 0
End"""

NONLINEAR_EXAMPLE = """Minimize
x1 + 2*x2 + 1.5 * x3 - 0.5*x4 - x5
Subject To
x1*x2 + 2*x3*x4 + 3 * x5*x6 - 5*x7*x8 - x9*x10 >= 1
End"""


def readers():
    from bdd_amd import native
    return [("python", parse_lp), ("c++", native.parse_lp)]


@pytest.mark.parametrize("which", [0, 1])
def test_reference_parser_vectors(which):
    name, read = readers()[which]
    # test_ILP_parser.cpp:37-57 — bounds that fix nothing leave the model alone
    ilp = read(ILP_EXAMPLE)
    assert ilp.var_names == ["x1", "x2", "x3", "x4", "x5"]
    assert ilp.objective == [1.0, 2.0, 1.5, -0.5, -1.0]
    assert len(ilp.constraints) == 1
    c = ilp.constraints[0]
    assert (c.coefficients, c.variables, c.ineq, c.rhs) == ([1, 2, 3, -5, -1], [0, 1, 2, 3, 4], ">=", 1)
    # :59-79 — `#` inside names, a row identifier with `;` and `@`
    ilp = read(ILP_EXAMPLE_HASH)
    assert ilp.var_names == ["x1", "x2", "x#3", "x#4", "x3"]
    assert ilp.objective == [1.0, 2.0, 1.5, -0.5, -1.0]
    assert len(ilp.constraints) == 1 and ilp.constraints[0].name == "b_cuta_;0;14_1;@41d"
    # :81-106 — products of variables: the reference builds monomials (convert_nonlinear_to_bdd); this reader has no
    # converter for them and must refuse, never read `x1*x2` as `x1 + x2`
    with pytest.raises(ValueError, match="nonlinear"):
        read(NONLINEAR_EXAMPLE)
    for row in ("x1 x2 + x3 >= 1", "x1 + 2 x2 * x3 >= 1", "x1 + x2 x3 >= 1"):
        with pytest.raises(ValueError, match="nonlinear"):
            read(f"Minimize\nx1 + x2 + x3\nSubject To\n{row}\nEnd\n")
    with pytest.raises(ValueError, match="sign"):
        read("Minimize\nx1 x2\nSubject To\nx1 + x2 >= 1\nEnd\n")


BOUNDS_BASE = "Minimize\n2 x1 + 3 x2 - x3 + 4 x4 + 1.5\nSubject To\nr1: x1 + x2 + x3 >= 1\nr2: x1 - x3 + 2 x4 <= 1\nr3: x2 + x4 = 1\n"


@pytest.mark.parametrize("which", [0, 1])
def test_bounds_fix_variables_as_ilp_input_reduce(which):
    """ILP_parser.cpp:128-131,343-436 (the four line forms) + ILP_input::reduce (ILP_input.cpp:508-591)."""
    name, read = readers()[which]
    ilp = read(BOUNDS_BASE + "Bounds\n x1 = 1\n x3 = 0\nBinaries\n x1 x2\nEnd\n")
    assert ilp.var_names == ["x2", "x4"] and ilp.objective == [3.0, 4.0]
    assert ilp.constant == 1.5 + 2.0
    got = [(c.name, c.coefficients, c.variables, c.ineq, c.rhs) for c in ilp.constraints]
    assert got == [("r1", [1], [0], ">=", 0), ("r2", [2], [1], "<=", 0), ("r3", [1, 1], [0, 1], "=", 1)]
    # every accepted way of writing the same two fixations
    for lines in (" x1 >= 1\n x3 <= 0\n", " 1 <= x1\n 0 >= x3\n", " 1 <= x1 <= 1\n 0 <= x3 <= 0\n", " 1 = x1\n x3 = 0\n x2 <= 1\n 0 <= x4 <= 1\n x4 >= 0\n"):
        again = read(BOUNDS_BASE + "Bounds\n" + lines + "End\n")
        assert again.var_names == ["x2", "x4"] and again.constant == 3.5
        assert [(c.name, c.coefficients, c.variables, c.ineq, c.rhs) for c in again.constraints] == got
    # a row that loses every term is dropped when it holds ...
    ilp = read(BOUNDS_BASE + "Bounds\n x2 = 1\n x4 = 0\nEnd\n")
    assert [c.name for c in ilp.constraints] == ["r1", "r2"] and ilp.var_names == ["x1", "x3"]
    assert [(c.coefficients, c.rhs) for c in ilp.constraints] == [([1, 1], 0), ([1, -1], 1)]
    # ... and is an error when it does not
    with pytest.raises(ValueError, match="not feasible.*r3"):
        read(BOUNDS_BASE + "Bounds\n x2 = 1\n x4 = 1\nEnd\n")
    with pytest.raises(ValueError, match="not feasible.*r3"):
        read(BOUNDS_BASE + "Bounds\n x2 = 0\n x4 = 0\nEnd\n")


@pytest.mark.parametrize("which", [0, 1])
@pytest.mark.parametrize("line,match", [("x1 = 2", "0 or 1"), ("x1 <= 0.5", "0 or 1"), ("nope = 1", "no row"), ("x1 free", "expected"),
                                        ("x1 < 1", "expected|0 or 1"), ("1 <= x1 <= 0", "above"), ("-1 <= x1", "0 or 1"), ("x1 = 0\n x1 >= 1", "0 and to 1"),
                                        ("0 <= x1 >= 1", "expected"),
                                        # malformed relations: no reader may skip the stray character and read a fixation
                                        ("x1 < = 1", "expected"), ("x1 <= 1 <", "expected"), ("x1 > = 0", "expected"), ("> x1 = 1", "expected")])
def test_bounds_lines_that_are_refused(which, line, match):
    name, read = readers()[which]
    with pytest.raises(ValueError, match=match):
        read(BOUNDS_BASE + "Bounds\n " + line + "\nEnd\n")


def test_fixation_equals_solving_the_restricted_model_by_enumeration():
    """reduce() keeps the optimum of the model restricted to the fixation (checked by brute force on a small instance)."""
    import itertools
    full = parse_lp(BOUNDS_BASE + "End\n")
    for fix in ({"x1": 1}, {"x3": 0}, {"x1": 1, "x3": 0}, {"x2": 0, "x4": 1}, {"x1": 0, "x2": 0}):
        red = parse_lp(BOUNDS_BASE + "Bounds\n" + "".join(f" {k} = {v}\n" for k, v in fix.items()) + "End\n")
        best_full = min((full.evaluate(x) for x in itertools.product((0, 1), repeat=4)
                         if full.feasible(x) and all(x[full.var_names.index(k)] == v for k, v in fix.items())), default=None)
        best_red = min((red.evaluate(x) for x in itertools.product((0, 1), repeat=red.nr_variables()) if red.feasible(x)), default=None)
        assert best_full == best_red, fix


def test_fix_variable_known_answers_on_the_two_oracles():
    """test/test_bdd_solver_fix_variable.cpp:6-48 with the fixations written as Bounds: 1, 2 and 3 with the CPU rule; the GPU rule
    (no exchange on a layer with a non-finite min-marginal, bdd_cuda_parallel_mma.cu:83-84) stops at 2.4853887... on the third,
    whose reduced model has a row that forces two variables to 0 — the value the GPU tests expect of the HIP solver."""
    import numpy as np
    from bdd_amd import to_bdd_collection
    from oracle.oracle import CudaRuleOracle, Oracle
    chain = ("Minimize\n3 mu_1_0 + 1 mu_1_1\n- 1 mu_2_0 + 0 mu_2_1\n+ 1 mu_00 + 2 mu_10 + 1 mu_01 + 0 mu_11\nSubject To\n"
             "mu_1_0 + mu_1_1 = 1\nmu_2_0 + mu_2_1 = 1\nmu_00 + mu_10 + mu_01 + mu_11 = 1\nmu_1_0 - mu_00 - mu_01 = 0\n"
             "mu_1_1 - mu_10 - mu_11 = 0\nmu_2_0 - mu_00 - mu_10 = 0\nmu_2_1 - mu_01 - mu_11 = 0\n")
    for bounds, want_cpu, want_gpu in (("", 1.0, 1.0), ("Bounds\n mu_2_1 = 0\n", 2.0, 2.0), ("Bounds\n mu_2_1 = 0\n mu_1_1 = 0\n", 3.0, 2.4853887428038)):
        ilp = parse_lp(chain + bounds + "End\n")
        col = to_bdd_collection(ilp)
        cpu, gpu = Oracle(col, np.array(ilp.objective), "double"), CudaRuleOracle(col, np.array(ilp.objective), "double")
        for _ in range(200):
            cpu.iteration()
            gpu.iteration(0.5)
        assert abs(cpu.lower_bound() + ilp.constant - want_cpu) <= 1e-6
        assert abs(gpu.lower_bound() + ilp.constant - want_gpu) <= 1e-6
