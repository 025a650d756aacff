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
