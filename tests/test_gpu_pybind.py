"""The pybind11 module bdd_solver_py (bdd_amd/csrc/host/bdd_solver_py.cpp) — needs an MI355X.

bdd_solver_py.bdd_solver binds the C++ driver bddmma_host::bdd_solver as the reference's module does
(src/bdd_solver/bdd_solver_py.cpp:9-20); bdd_solver_py.bdd_hip_parallel_mma is the GPU solver class of
src/bdd_solver/bdd_cuda_parallel_mma_py.cu:15-80 with pickle through the solver's own archive.  The known answers are those of
the reference's end-to-end tests; every result is also compared with the Python driver bdd_amd.bdd_solver on the same config."""
import pickle

import numpy as np
import pytest

from bdd_amd import bdd_solver_py
from bdd_amd.bdd_solver import bdd_solver as py_driver
from bdd_amd.instances import GRID_3X3, LONG_CHAIN, assignment_ilp, mrf_ilp, random_set_cover
from bdd_amd.solver import bdd_hip_parallel_mma

pytestmark = pytest.mark.gpu

TC = {"maximum iterations": 200, "improvement slope": 0.0, "minimum improvement": 0.0, "time limit": 1e10}


def cfg(lp, **kw):
    c = {"precision": "double", "relaxation solver": "cuda parallel mma", "termination criteria": dict(TC), "input": lp}
    c.update(kw)
    return c


def test_driver_kats_from_dict_and_from_json_text():
    import json
    s = bdd_solver_py.bdd_solver(cfg(assignment_ilp(3).write_lp()), quiet=True).solve()      # ctor from a dict
    assert abs(s.lower_bound() - (-6.0)) <= 1e-6                                             # test_bdd_bipartite_matching_problem.cpp:8-59
    c = -np.ones((3, 3)); c[:, 0] = -2
    s = bdd_solver_py.bdd_solver(json.dumps(cfg(assignment_ilp(3, c).write_lp())), True).solve()   # ctor from JSON text
    assert abs(s.lower_bound() - (-4.0)) <= 1e-6
    assert s.result["iterations"] <= 200 and s.result["lb_final"] >= s.result["lb_initial"] - 1e-9
    assert s.solution is None                                                                 # no rounding requested
    base = "Minimize\nx1 + x2 + x3 + x4 + x5 + x6\nSubject To\nx1 + x2 + x4 >= 1\nx1 + x3 + x5 >= 1\nx2 + x3 + x6 >= 1\n"
    tail = "Bounds\nBinaries\nx1\nx2\nx3\nx4\nx5\nx6\nEnd\n"
    s = bdd_solver_py.bdd_solver(cfg(base + tail), quiet=True).solve()
    assert abs(s.lower_bound() - 1.5) <= 1e-6                                                 # test_loose_covering_problem.cpp:59
    with pytest.raises(RuntimeError):
        bdd_solver_py.bdd_solver(cfg(base + tail, **{"relaxation solver": "sequential mma"}), quiet=True).solve()


@pytest.mark.parametrize("solver", ["cuda parallel mma", "lbfgs cuda mma"])
def test_driver_agrees_with_the_python_driver(solver):
    ilp = mrf_ilp(**LONG_CHAIN)
    c = cfg(ilp.write_lp(), **{"relaxation solver": solver,
                               "perturbation rounding": {"initial perturbation": 0.1, "perturbation growth rate": 1.2,
                                                         "inner iterations": 50, "outer iterations": 60}})
    a = bdd_solver_py.bdd_solver(c, quiet=True).solve()
    b = py_driver(c, quiet=True).solve()
    assert abs(a.lower_bound() - b.lower_bound()) <= 1e-9 * max(1.0, abs(b.lower_bound()))
    assert abs(a.result["lb_final"] - (-9.0)) < 1e-6                                          # test_bdd_cuda_parallel_mma.cu:230
    assert a.solution is not None and ilp.feasible(a.solution) and abs(ilp.evaluate(a.solution) - (-9.0)) < 1e-9
    assert abs(a.solution_objective - ilp.evaluate(a.solution)) < 1e-12
    mm_a, mm_b = a.min_marginals(), b.min_marginals()
    assert len(mm_a) == len(mm_b) and all(np.asarray(x).reshape(-1, 2).shape == y.shape for x, y in zip(mm_a, mm_b))
    if solver == "cuda parallel mma":   # (two L-BFGS + random-perturbation runs end in different reparametrisations of the same optimum)
        for x, y in zip(mm_a, mm_b):
            np.testing.assert_allclose(np.asarray(x).reshape(-1, 2), y, atol=1e-6)
    names, m0, m1 = a.min_marginals_with_variable_names()
    nb, b0, b1 = b.min_marginals_with_variable_names()
    assert list(names) == list(nb) and [len(p) for p in m0] == [len(q) for q in b0] and [len(p) for p in m1] == [len(q) for q in b1]


def test_gpu_solver_class_pickle_and_min_marginal_diff():
    torch = pytest.importorskip("torch")
    ilp = mrf_ilp(**GRID_3X3)
    s = bdd_solver_py.bdd_hip_parallel_mma(ilp.write_lp(), precision="float")
    assert "nr_variables" in repr(s)
    assert s.nr_primal_variables() == ilp.nr_variables() and s.nr_bdds() == len(ilp.constraints)
    s.iterations(30)
    lb = s.lower_bound()
    t = pickle.loads(pickle.dumps(s))                       # bdd_cuda_parallel_mma_py.cu:15-37
    assert (t.nr_layers(), t.nr_hops(), t.nr_bdds()) == (s.nr_layers(), s.nr_hops(), s.nr_bdds())
    # nr_layers(hop_index), bdd_cuda_parallel_mma_py.cu:53: the per-hop counts add up to the total
    assert sum(s.nr_layers(h) for h in range(s.nr_hops())) == s.nr_layers() and s.nr_layers(0) == s.nr_bdds()
    with pytest.raises(IndexError):
        s.nr_layers(s.nr_hops())
    assert t.lower_bound() == lb
    s.iteration(); t.iteration()
    assert abs(t.lower_bound() - s.lower_bound()) <= 1e-6 * max(1.0, abs(s.lower_bound()))
    # compute_and_set_min_marginal_diff writes hi - lo per layer into memory allocated by Python (:56-72)
    buf = torch.empty(s.nr_layers(), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    s.compute_and_set_min_marginal_diff(buf.data_ptr())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(buf.cpu().numpy(), np.asarray(s.min_marginal_diff(), np.float32))
    # and against the ctypes mirror on the same problem
    from bdd_amd import parse_lp, to_bdd_collection
    p = parse_lp(ilp.write_lp())
    ref = bdd_hip_parallel_mma(to_bdd_collection(p), p.objective, precision="float")
    ref.iterations(31)
    _, mm0, mm1 = ref.min_marginals_cuda(get_sorted=False)
    np.testing.assert_allclose(ref.min_marginal_diff(), mm1 - mm0, atol=1e-6)
    assert abs(ref.lower_bound() - s.lower_bound()) <= 1e-5 * max(1.0, abs(s.lower_bound()))


def test_objective_only_variables_at_any_index():
    """a variable that occurs in the objective only is free wherever its index lies.  The LP reader numbers variables by
    first appearance and reads the objective first, so `y` below gets index 1 — between constrained variables.  Its better value
    enters the bound (min(0, c)) and the rounded solution (c < 0 -> 1) in both drivers and in the pybind solver class."""
    lp = ("Minimize\nx1 - 3 y + 1.5 x2 + 2 x3 - 0.5 z + 4 w + 2 x4 - x5\nSubject To\nx1 + x2 + x3 >= 1\nx4 + x5 = 1\n"
          "Bounds\nBinaries\nx1\ny\nx2\nx3\nz\nw\nx4\nx5\nEnd\n")
    from bdd_amd import parse_lp
    ilp = parse_lp(lp)
    names = list(ilp.var_names)
    assert names.index("y") == 1 and names.index("z") == 4 and names.index("w") == 5
    c = cfg(lp, **{"perturbation rounding": {"initial perturbation": 0.1, "perturbation growth rate": 1.2, "inner iterations": 20,
                                             "outer iterations": 20}})
    want = 1.0 - 1.0 - 3.0 - 0.5   # x1 = 1, x5 = 1; y = 1, z = 1, w = 0 are free
    for driver in (lambda: bdd_solver_py.bdd_solver(c, quiet=True).solve(), lambda: py_driver(c, quiet=True).solve()):
        s = driver()
        assert abs(s.lower_bound() - want) <= 1e-9
        sol = list(s.solution)
        assert ilp.feasible(sol) and abs(ilp.evaluate(sol) - want) <= 1e-12
        assert sol[1] == 1 and sol[4] == 1 and sol[5] == 0
    g = bdd_solver_py.bdd_hip_parallel_mma(lp, precision="double")
    assert abs(g.constant - (-3.5)) <= 1e-12
    g.iterations(20)
    assert abs(g.lower_bound() - want) <= 1e-9
    t = pickle.loads(pickle.dumps(g))
    assert t.constant == g.constant and t.device() == g.device() and t.lower_bound() == g.lower_bound()
