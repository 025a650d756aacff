"""JSON config surface of the bdd_solver driver — CPU-only part (parsing, option validation, error behaviour)."""
import json

import pytest

from bdd_amd.bdd_solver import bdd_solver
from bdd_amd.instances import assignment_ilp

LP = assignment_ilp(3).write_lp()


def test_config_from_dict_string_and_file(tmp_path):
    cfg = {"input": LP, "relaxation solver": "cuda parallel mma"}
    assert bdd_solver(cfg).config == cfg
    assert bdd_solver(json.dumps(cfg)).config == cfg
    p = tmp_path / "c.json"
    p.write_text(json.dumps(cfg))
    assert bdd_solver(str(p)).config == cfg


def test_input_file_or_string(tmp_path):
    p = tmp_path / "m.lp"
    p.write_text(LP)
    a = bdd_solver({"input": str(p)}, quiet=True).read_ILP()
    b = bdd_solver({"input": LP}, quiet=True).read_ILP()
    assert a.var_names == b.var_names and a.objective == b.objective
    with pytest.raises(RuntimeError, match="no input specified"):
        bdd_solver({}, quiet=True).read_ILP()


def test_option_errors_match_reference_behaviour():
    s = bdd_solver({"input": LP, "variable order": "spiral"}, quiet=True)
    with pytest.raises(RuntimeError, match="Variable order spiral unknown"):   # bdd_solver.cpp:93
        s.process_ILP(s.read_ILP())
    s = bdd_solver({"input": LP, "relaxation solver": "quantum mma"}, quiet=True)
    ilp = s.read_ILP()
    col = s.transform_to_BDDs(ilp)
    with pytest.raises(RuntimeError, match="relaxation solver quantum mma unknown"):  # :265
        s.construct_solver(col, ilp.objective)
    s = bdd_solver({"input": LP, "relaxation solver": "sequential mma"}, quiet=True)
    with pytest.raises(RuntimeError, match="CPU solver"):
        s.construct_solver(col, ilp.objective)
    s = bdd_solver({"input": LP, "precision": "half"}, quiet=True)
    with pytest.raises(RuntimeError, match="precision must be"):                # :142-143
        s.construct_solver(col, ilp.objective)


def test_normalize_constraints_sorts_monomials():
    lp = "Minimize\nc + b + a\nSubject To\na + c + b >= 1\nEnd\n"
    s = bdd_solver({"input": lp, "normalize constraints": True}, quiet=True)
    ilp = s.read_ILP()
    assert ilp.constraints[0].variables == [2, 0, 1]
    s.process_ILP(ilp)
    assert ilp.constraints[0].variables == [0, 1, 2]
