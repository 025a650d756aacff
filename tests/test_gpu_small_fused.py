"""Instances that fit one workgroup: whole iterations inside one launch (bdd_amd/csrc/kernels/small.hpp: k_iterate_small) against
the four-launches-per-iteration path of the same library (variant_flags bit 19), the CPU oracle and the reference's known answers.

The fused kernel runs the hop loops of the resident sweeps on the same records (same operations, same order) and an exchange that sums a
variable's differences in (variable, bdd) order in double and rounds once — so it is BIT-EQUAL to the sequential launches for float
instances (the double sums of a few floats are exact) and for any instance whose variables sit in at most two BDDs (a + b commutes), and
within rounding of the LDS atomics' order otherwise.  run_solver's tests run inside the kernel after every iteration
(include/run_solver_util.h:40-73)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bdd_amd import to_bdd_collection  # noqa: E402
from bdd_amd.instances import assignment_ilp, random_set_cover  # noqa: E402
from bdd_amd.solver import bdd_hip_parallel_mma, run_solver  # noqa: E402
from oracle.oracle import Oracle  # noqa: E402

SEQ = 0x80000   # variant_flags bit 19: four launches per iteration


def matching(n, precision, costs=None, **kw):
    ilp = assignment_ilp(n, costs)
    s = bdd_hip_parallel_mma(to_bdd_collection(ilp), ilp.objective, precision=precision, **kw)
    return s, ilp


def same_state(a, b, exact=True, rel=0.0):
    for x, y in zip(a.get_solver_costs(), b.get_solver_costs()):
        if exact:
            np.testing.assert_array_equal(x, y)
        else:
            np.testing.assert_allclose(x, y, rtol=rel, atol=rel)
    la, lb = a.lower_bound(), b.lower_bound()
    assert la == lb if exact else abs(la - lb) <= rel * max(1.0, abs(lb))


@pytest.mark.parametrize("precision", ["float", "double"])
@pytest.mark.parametrize("n", [3, 8])
def test_assignment_problem_runs_fused_and_equals_the_sequential_launches_bit_for_bit(n, precision):
    """BASELINE.json configs[0]: n x n bipartite matching (test_bdd_bipartite_matching_problem.cpp:8-59).  Every variable sits in two BDDs."""
    f, ilp = matching(n, precision)
    q, _ = matching(n, precision, variant_flags=SEQ)
    assert f.fused_small() and not q.fused_small()
    assert f.solve_sweep_kind() == q.solve_sweep_kind() == "resident2"
    assert f.lower_bound() == q.lower_bound()
    for k in (1, 2, 5, 17):          # launches of k iterations against k x four launches
        f.iterations(k)
        q.iterations(k)
        same_state(f, q)
    f.iteration(); q.iteration()     # a single iteration is a launch of one
    same_state(f, q)
    for mf, mq in zip(f.min_marginals(), q.min_marginals()):
        np.testing.assert_array_equal(mf, mq)
    # the reference's known answers: optimum -2 n, reached by the relaxation
    f.iterations(200)
    assert abs(f.lower_bound() - (-2.0 * n)) <= (1e-4 if precision == "float" else 1e-9)


@pytest.mark.parametrize("rows,k,fused", [(20, 6, ("float", "double")), (60, 5, ("float", "double")), (100, 7, ("float", "double")), (220, 8, ("float", "double")),
                                          (300, 9, ("float",)), (500, 12, ())],
                         ids=["1pack", "2packs", "4packs", "7packs", "10packs_records_in_L2", "16packs_beyond_one_CUs_LDS"])
def test_small_set_cover_float_is_bit_equal_and_double_within_rounding(rows, k, fused):
    """`fused`: the precisions whose state fits one CU's LDS (64 slots per hop and pack of potentials, twice, + costs, staging area, tables):
    the rule is capacity, not a pack count — the last cases keep their four launches, in double or altogether."""
    col, costs = random_set_cover(max(40, rows * 2 // 3), rows, k, seed=rows)
    for precision in ("float", "double"):
        f = bdd_hip_parallel_mma(col, costs, precision=precision)
        q = bdd_hip_parallel_mma(col, costs, precision=precision, variant_flags=SEQ)
        assert f.fused_small() == (precision in fused) and f.nr_packs() <= 16, (f.nr_packs(), f.solve_sweep_kind())
        f.iterations(12)
        q.iterations(12)
        f.iterations(3)
        q.iterations(3)
        if precision == "float":
            same_state(f, q)
        else:
            same_state(f, q, exact=False, rel=1e-12)
        o = Oracle(col, costs, precision)
        for _ in range(15):
            o.iteration()
        ref = o.lower_bound()
        assert abs(f.lower_bound() - ref) <= (1e-5 if precision == "float" else 1e-9) * abs(ref)


def test_state_written_back_is_what_every_other_entry_point_expects():
    """After a fused launch the potentials, arc costs, deferred differences and pending delta pairs in global memory are the solver's state:
    explicit passes, cost updates and more fused launches continue from it like the sequential path."""
    col, costs = random_set_cover(120, 150, 6, seed=9)
    f = bdd_hip_parallel_mma(col, costs, precision="float")
    q = bdd_hip_parallel_mma(col, costs, precision="float", variant_flags=SEQ)
    assert f.fused_small()
    rng = np.random.default_rng(3)
    for step in range(4):
        f.iterations(5)
        q.iterations(5)
        same_state(f, q)
        d = rng.uniform(-0.5, 0.5, size=f.nr_variables())
        f.update_costs([], d)
        q.update_costs([], d)
        np.testing.assert_array_equal(f.bdds_solution_vec(), q.bdds_solution_vec())
        np.testing.assert_array_equal(f.lower_bound_per_bdd(), q.lower_bound_per_bdd())
    for mf, mq in zip(f.min_marginals(), q.min_marginals()):
        np.testing.assert_array_equal(mf, mq)
    # distribute_delta folds the pending pairs and deferred differences into the arc costs: both must have been written back
    f.distribute_delta(); q.distribute_delta()
    same_state(f, q)


@pytest.mark.parametrize("precision", ["float", "double"])
@pytest.mark.parametrize("case", [dict(max_iter=23, tolerance=0.0, slope=0.0), dict(max_iter=1000, tolerance=1e-4, slope=0.0),
                                  dict(max_iter=1000, tolerance=0.0, slope=0.02), dict(max_iter=1000, tolerance=1e-6, slope=1e-9),
                                  dict(max_iter=64, tolerance=0.0, slope=0.0), dict(max_iter=129, tolerance=0.0, slope=0.0)],
                         ids=["max_iter", "tolerance", "slope", "defaults", "one_chunk", "two_chunks_and_one"])
def test_run_solver_with_the_tests_inside_the_kernel(precision, case):
    """bddmma_run_solver on a fused instance: chunks of 64 iterations per launch, the criteria tested in the kernel after each; iteration
    count, stop reason, bounds and the state left behind equal the sequential launches' (which the other run_solver tests pin on the
    reference's loop)."""
    # an assignment problem with random costs: the bound moves for a few dozen iterations, and every variable sits in two BDDs (bit-equal in double too)
    costs = np.random.default_rng(8).uniform(-3.0, 1.0, size=(12, 12))
    f, _ = matching(12, precision, costs)
    q, _ = matching(12, precision, costs, variant_flags=SEQ)
    assert f.fused_small() and not q.fused_small()
    rf = run_solver(f, max_iter=case["max_iter"], tolerance=case["tolerance"], improvement_slope=case["slope"], time_limit=1e9)
    rq = run_solver(q, max_iter=case["max_iter"], tolerance=case["tolerance"], improvement_slope=case["slope"], time_limit=1e9)
    assert (rf["iterations"], rf["stop_reason"]) == (rq["iterations"], rq["stop_reason"])
    assert rf["lb_initial"] == rq["lb_initial"] and rf["lb_final"] == rq["lb_final"]
    if case["tolerance"] or case["slope"]:
        assert rf["stop_reason"] in (2, 3) and rf["iterations"] < case["max_iter"]
    same_state(f, q)     # nothing ran behind the iteration that met the criterion
    f.iteration(); q.iteration()
    same_state(f, q)


def test_run_solver_reaches_the_reference_known_answers():
    for n, want in ((3, -6.0), (8, -16.0)):
        s, _ = matching(n, "double")
        assert s.fused_small()
        res = run_solver(s, max_iter=500, tolerance=1e-9, improvement_slope=0.0, time_limit=1e9)
        assert abs(res["lb_final"] - want) <= 1e-6 and res["iterations"] < 500
        assert s.lower_bound() == res["lb_final"]


def test_profiling_and_lbfgs_fall_back_to_the_launch_per_pass_path():
    from bdd_amd.solver import bdd_hip_lbfgs
    f, _ = matching(8, "double")
    q, _ = matching(8, "double", variant_flags=SEQ)
    f.set_profiling(True, stride=1)
    f.iterations(4)
    p = f.get_profile()
    assert p["launches"][0] == 4 and p["launches"][1] == 4   # event pairs need the launches
    f.set_profiling(False)
    q.iterations(4)
    same_state(f, q)
    lf, lq = bdd_hip_lbfgs(f), bdd_hip_lbfgs(q)
    for _ in range(8):
        lf.iteration(); lq.iteration()
    same_state(f, q)


@pytest.mark.parametrize("opts", [dict(waves_per_block=2), dict(waves_per_block=4), dict(waves_per_block=8), dict(vars_per_bin=64), dict(keep_bdd_order=True),
                                  dict(stage_cap=128), dict(pack_width=128), dict(resident_sweeps=1), dict(deterministic=True), dict(exchange_by_variable=2)],
                         ids=lambda o: ",".join(f"{k}={v}" for k, v in o.items()))
def test_layout_options_on_a_fused_instance(opts):
    """The staging tables' LDS slots are quad-relative (packs per workgroup of the four-launch path): with 2 / 4 / 8 packs per quad the fused kernel's
    prologue and epilogue map them to its own per-pack staging area.  Options under which the preconditions do not hold (several stage groups
    per pack, packs of 128 slots, no resident records, the deterministic / by-variable exchanges) must fall back, not misbehave."""
    col, costs = random_set_cover(146, 220, 8, seed=220)
    f = bdd_hip_parallel_mma(col, costs, precision="float", **opts)
    q = bdd_hip_parallel_mma(col, costs, precision="float", variant_flags=SEQ, **opts)
    fused_expected = not any(k in opts for k in ("stage_cap", "pack_width", "resident_sweeps", "deterministic", "exchange_by_variable"))
    assert f.fused_small() == fused_expected, opts
    for k in (3, 8):
        f.iterations(k)
        q.iterations(k)
        same_state(f, q)
    rf = run_solver(f, max_iter=70, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)
    rq = run_solver(q, max_iter=70, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)
    assert rf["iterations"] == rq["iterations"] == 70 and rf["lb_final"] == rq["lb_final"]
    same_state(f, q)
