"""L-BFGS and primal-rounding parity — needs an MI355X.

L-BFGS: bddmma_lbfgs_iteration against the CPU restatement oracle/lbfgs_oracle.py — same solver
selection, same number of step-size trials, same step sizes, same lower-bound trajectory (1e-9 rel. in double; float is
held to 1e-5 rel. on the bound while the decision sequence agrees).  Includes BASELINE.json configs[3]: L-BFGS on the
10.5 M-node instance.

Rounding (§8 f-1): one perturb_primal_costs round against the numpy restatement oracle/rounding_oracle.py — identical
#one / #zero / #equal / #inconsistent, identical cost deltas for the deterministic types, and the resulting arc costs.
"""
import os

import numpy as np
import pytest

from bdd_amd.instances import assignment_ilp, mrf_ilp, GRID_3X3, random_set_cover
from bdd_amd import to_bdd_collection
from bdd_amd.solver import bdd_hip_lbfgs, bdd_hip_parallel_mma
from oracle import rounding_oracle as R
from oracle.lbfgs_oracle import LbfgsOracle
from oracle.oracle import Oracle

pytestmark = pytest.mark.gpu


def run_lbfgs_pair(col, costs, precision, iters, threads=1, solver_options=None, between=None, **params):
    s = bdd_hip_parallel_mma(col, costs, precision=precision, **(solver_options or {}))
    l = bdd_hip_lbfgs(s, **params)
    o = LbfgsOracle(Oracle(col, costs, precision, threads=threads), **params)
    # In double everything is compared, to 1e-9, for all iterations.  Float: the subgradient (an argmin path per BDD) is
    # discontinuous in the costs, so float rounding differences between the two implementations are amplified by every
    # L-BFGS step (observed: 1e-6 rel. after 30 iterations, 1e-4 after 50); the bound is held to 1e-4 rel. for the first 30
    # iterations with the same mma / lbfgs choices, afterwards to 2e-3 rel. and to monotonicity (lbfgs_impl.h:403).
    exact = precision == "double"
    n_lbfgs = 0
    prev_lb = -np.inf
    for it in range(iters):
        if between is not None and between(it, s, l, o):   # something done to both sides between two iterations; True: the costs have changed
            prev_lb = -np.inf
        l.iteration()
        o.iteration()
        st = l.state()
        lb, ref = l.lower_bound(), o.lower_bound()
        ctx = (it, precision, st, o.last_kind, o.last_trials, o.step_size)
        rel = 1e-9 if exact else (1e-4 if it < 30 else 2e-3)
        assert abs(lb - ref) <= rel * max(1.0, abs(ref)), ctx
        assert lb >= prev_lb - 1e-5 * abs(lb), ctx
        prev_lb = lb
        if not exact and it >= 30:
            n_lbfgs += st["last_kind"]
            continue
        # the state machine took the same decisions: mma / lbfgs, how many trial steps, which step stayed applied
        assert st["last_kind"] == o.last_kind, ctx
        n_lbfgs += st["last_kind"]
        if not exact:
            continue
        assert st["last_trials"] == o.last_trials, ctx
        assert st["history_entries"] == len(o.history), ctx
        assert st["num_unsuccessful_updates"] == o.num_unsuccessful, ctx
        assert abs(st["step_size"] - o.step_size) <= 1e-12 * o.step_size, ctx
        assert abs(st["last_applied_step"] - o.last_applied_step) <= 1e-12 * max(o.last_applied_step, 1e-300), ctx
    if exact:
        assert (st["mma_iterations"], st["lbfgs_iterations"]) == (o.mma_iterations, o.lbfgs_iterations)
    assert st["mma_iterations"] + st["lbfgs_iterations"] == iters
    return n_lbfgs, lb


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("seed,nv,nr,k", [(13, 3000, 2500, 8), (7, 1200, 1500, 5)])
def test_lbfgs_trajectory_vs_oracle(precision, seed, nv, nr, k):
    col, costs = random_set_cover(nv, nr, k, seed=seed)
    n_lbfgs, _ = run_lbfgs_pair(col, costs, precision, 60)
    assert n_lbfgs >= 20   # the comparison did cover L-BFGS steps, not only the mma iterations that collect the history


def test_lbfgs_other_parameters_vs_oracle():
    col, costs = random_set_cover(2000, 1800, 6, seed=21)
    n_lbfgs, _ = run_lbfgs_pair(col, costs, "double", 40, history_size=3, init_step_size=1e-4, req_rel_lb_increase=1e-3,
                                step_size_decrease_factor=0.5, step_size_increase_factor=1.5)
    assert n_lbfgs >= 10


def test_lbfgs_structured_instance_vs_oracle():
    ilp = mrf_ilp(**GRID_3X3)
    col = to_bdd_collection(ilp)
    run_lbfgs_pair(col, np.asarray(ilp.objective, float), "double", 40)


STAGED, GATHER = 0x200, 0x400   # bddmma_options.variant_flags bits 9 / 10: make_dual_feasible(direction) through the staging tables / by gathers


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("options", [dict(), dict(waves_per_block=1, pack_width=64), dict(waves_per_block=8, stage_cap=128, vars_per_bin=64),
                                     dict(waves_per_block=2, vars_per_bin=2048), dict(vars_per_bin=4096), dict(exchange_by_variable=2)])
def test_lbfgs_staged_projection_vs_oracle(precision, options):
    """The projection of the direction as the large instances do it (k_stage_transpose / k_project_entries: layers -> entries through the
    sweeps' staging tables, per-variable means in LDS per exchange bin, back with the first step applied), forced on a small instance
    for every shape of the tables: 1 / 2 / 4 / 8 packs per workgroup, 256- / 512- / 1024-thread bins, several rounds per quad."""
    col, costs = random_set_cover(3000 if options.get("vars_per_bin", 0) < 2048 else 20000, 2500, 8, seed=5)
    n_lbfgs, _ = run_lbfgs_pair(col, costs, precision, 40, solver_options=dict(variant_flags=STAGED, **options))
    assert n_lbfgs >= 10


UNFUSED = 0x8000   # variant_flags bit 15: the direction as its own pass (k_lb_direction) instead of inside the projection's first pass


@pytest.mark.parametrize("flags", [0, STAGED])
def test_lbfgs_rejected_trial_steps_vs_oracle(flags):
    """A large initial step and a demanding required increase: most first trial steps are rejected, several trials per iteration, some
    iterations end without a step (the branches of lbfgs_impl.h:186-215 the default parameters rarely take)."""
    col, costs = random_set_cover(3000, 2500, 8, seed=29)
    s_opts = dict(variant_flags=flags)
    n_lbfgs, _ = run_lbfgs_pair(col, costs, "double", 45, solver_options=s_opts, init_step_size=1e-2, req_rel_lb_increase=5e-2,
                                step_size_decrease_factor=0.6, step_size_increase_factor=1.3)
    assert n_lbfgs >= 10


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("flags,params,options", [(STAGED | UNFUSED, {}, {}), (STAGED, dict(history_size=3), {}), (STAGED, dict(history_size=8), dict(waves_per_block=2)),
                                                  (STAGED | UNFUSED, dict(history_size=3), dict(waves_per_block=1, pack_width=64)),
                                                  (STAGED, dict(history_size=2), dict(waves_per_block=8, stage_cap=128, vars_per_bin=64))])
def test_lbfgs_direction_inside_the_projection_vs_oracle(precision, flags, params, options):
    """Where the projection is staged and all packs are narrow the direction is formed inside its layers -> entries pass straight from the
    history (k_stage_lincomb; test_lbfgs_staged_projection_vs_oracle runs that path with the default history of five): other history sizes
    (the run-time form of the kernel), and the separate direction pass, which instances with wide packs and small ones still take."""
    col, costs = random_set_cover(3000, 2500, 8, seed=11)
    n_lbfgs, _ = run_lbfgs_pair(col, costs, precision, 40, solver_options=dict(variant_flags=flags, **options), **params)
    assert n_lbfgs >= 10


@pytest.mark.parametrize("flags", [0, STAGED])
def test_lbfgs_sees_cost_changes_between_iterations(flags):
    """x = hi - lo + deferred mm reaches the wrapper as a view the backward solve sweep writes (SolverBase::lbfgs_views); whatever else
    changes costs between two iterations — update_costs through the wrapper (flushes the history, lbfgs_impl.h:343-364), update_costs /
    set_cost on the solver itself, distribute_delta — must make the next iteration rebuild it."""
    col, costs = random_set_cover(2500, 2000, 7, seed=3)
    rng = np.random.Generator(np.random.PCG64(17))
    nv = col.nr_variables()

    def between(it, s, l, o):
        if it == 14:      # through the wrapper: history flushed on both sides
            d = rng.uniform(-0.2, 0.3, nv).round(4)
            l.update_costs(np.zeros(0), d)
            o.update_costs(np.zeros(0), d)
        elif it == 27:    # behind the wrapper's back: the history stays, x must still be the current one
            d = rng.uniform(-0.05, 0.05, nv).round(4)
            s.update_costs(np.zeros(0), d)
            o.s.update_costs(np.zeros(0), d)
        elif it == 33:
            s.set_cost(0.125, 5)
            d = np.zeros(nv); d[5] = 0.125
            o.s.update_costs(np.zeros(0), d)
        return it in (14, 27, 33)

    n_lbfgs, _ = run_lbfgs_pair(col, costs, "double", 45, solver_options=dict(variant_flags=flags), between=between)
    assert n_lbfgs >= 10


def test_lbfgs_staged_projection_with_wide_and_long_packs():
    """Layers of wide packs have no staging tables (they go through lpos); long BDDs have several stage groups per pack."""
    from bdd_amd import BddCollection
    rng = np.random.Generator(np.random.PCG64(5))
    V = 400
    col = BddCollection()
    for _ in range(8):
        k = int(rng.integers(16, 22))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        co = rng.integers(1, 40, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), vs)
    col.add_covering(np.sort(rng.choice(V, size=200, replace=False)))
    for _ in range(300):
        col.add_covering(np.sort(rng.choice(V, size=6, replace=False)))
    costs = rng.uniform(0.5, 3, col.nr_variables()).round(3)
    for flags in (STAGED, GATHER):
        run_lbfgs_pair(col, costs, "double", 40, solver_options=dict(variant_flags=flags, pack_width=64, wide_pack_width=512, stage_cap=64))


@pytest.mark.parametrize("flags", [0, STAGED])
@pytest.mark.parametrize("precision,tol", [("double", 1e-9), ("float", 2e-3)])
def test_applied_direction_is_dual_feasible(precision, tol, flags):
    """make_dual_feasible(direction) (bdd_cuda_base.cu:1261-1303) keeps the cost of every variable: sum over its layers of hi - lo (with the
    deferred min-marginal differences distributed) stays the objective coefficient — after many L-BFGS steps the primal objective vector
    must still be the input costs."""
    col, costs = random_set_cover(3000, 2500, 8, seed=13)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, variant_flags=flags)
    l = bdd_hip_lbfgs(s)
    steps = 0
    for _ in range(80):
        l.iteration()
        steps += l.state()["last_kind"]
    assert steps >= 40                      # most iterations did apply an L-BFGS step
    s.distribute_delta()
    np.testing.assert_allclose(s.get_primal_objective_vector_host(), costs, rtol=0, atol=tol)


def test_lbfgs_full_size_config4():
    """BASELINE.json configs[3]: lbfgs parallel mma on "the same 10M-node instance" as configs[2] — the mt19937_64(12345) instance of
    bench.py and of the full-size fixture (defaults m = 5, step 1e-6, 1e-6, 0.8, 1.1)."""
    from bdd_amd.instances import random_set_cover_mt
    col, costs = random_set_cover_mt(1_000_000, 500_000, 10, 12345)
    assert col.nr_bdd_nodes() == 10_500_000
    n_lbfgs, lb = run_lbfgs_pair(col, costs, "double", 20, threads=min(os.cpu_count() or 1, 32))
    assert n_lbfgs >= 10
    # and in float against the double run: same bound to 1e-5 rel. after the same 20 iterations
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    l = bdd_hip_lbfgs(s)
    prev = s.lower_bound()
    for _ in range(20):
        l.iteration()
        cur = l.lower_bound()
        assert cur >= prev - 1e-5 * abs(prev)
        prev = cur
    assert abs(prev - lb) <= 1e-4 * abs(lb)   # float L-BFGS may take different trial steps; the bound still tracks double


# ------------------------------------------------------------------------------------------------ rounding
def numpy_round(s, delta):
    """the reference's perturb_primal_costs on the solver's own sorted min-marginals (the solver must have no deferred delta)"""
    var, mm0, mm1 = s.min_marginals_cuda(get_sorted=True)
    n = s.nr_variables()
    t = R.compute_mm_types(n, var, mm0, mm1)
    s0, s1 = R.compute_mm_sums(n, var, mm0, mm1, s.value_type)
    c0, c1, side = R.perturbation(t, s0, s1, delta, s.value_type)
    return t, c0, c1, side


def check_round(s, res, t, c0, c1, side, delta, lo_before, hi_before):
    assert res["counts"] == R.counts(t)
    det = side == -2
    np.testing.assert_array_equal(res["cost_delta_0"][det], c0[det])
    np.testing.assert_array_equal(res["cost_delta_1"][det], c1[det])
    rnd = ~det
    g0, g1 = res["cost_delta_0"][rnd], res["cost_delta_1"][rnd]
    assert np.all((g0 == 0) | (g1 == 0))                       # exactly one side is perturbed ...
    assert np.all(g0 + g1 <= delta * delta * (1 + 1e-6))       # ... by |r| * delta with |r| <= delta
    fixed = side >= 0                                           # inconsistent: side given by the sums (mm_0 < mm_1 -> 1)
    assert np.all(res["cost_delta_1"][fixed & (side == 1)] >= 0) and np.all(res["cost_delta_0"][fixed & (side == 1)] == 0)
    assert np.all(res["cost_delta_1"][fixed & (side == 0)] == 0)
    # update_costs(cost_delta_0, cost_delta_1): every layer of a variable gets delta / nr_bdds(var) (bdd_cuda_base.cu:457-474)
    lo, hi, _ = s.get_solver_costs()
    v = s.get_primal_variable_index()
    nb = s.get_num_bdds_per_var().astype(np.float64)
    exp_lo = (lo_before.astype(np.float64) + res["cost_delta_0"].astype(np.float64)[v] / nb[v]).astype(s.value_type)
    exp_hi = (hi_before.astype(np.float64) + res["cost_delta_1"].astype(np.float64)[v] / nb[v]).astype(s.value_type)
    np.testing.assert_array_equal(lo, exp_lo)
    np.testing.assert_array_equal(hi, exp_hi)


@pytest.mark.parametrize("precision", ["double", "float"])
def test_perturb_primal_costs_vs_numpy(precision):
    col, costs = random_set_cover(3000, 2500, 8, seed=31)
    s = bdd_hip_parallel_mma(col, costs, precision=precision)
    s.iterations(25)
    s.distribute_delta()
    delta = 0.3
    t, c0, c1, side = numpy_round(s, delta)
    assert min(R.counts(t)) > 0 or R.counts(t)[3] > 0     # the instance exercises several types
    lo, hi, _ = s.get_solver_costs()
    res = s.perturb_primal_costs(delta, round_index=0, seed=5)
    check_round(s, res, t, c0, c1, side, delta, lo, hi)
    # different round / seed: same classification on the same state is not expected (costs moved), but the deterministic
    # table holds again
    s.distribute_delta()
    t, c0, c1, side = numpy_round(s, 2 * delta)
    lo, hi, _ = s.get_solver_costs()
    res2 = s.perturb_primal_costs(2 * delta, round_index=1, seed=5)
    check_round(s, res2, t, c0, c1, side, 2 * delta, lo, hi)


def test_classification_from_the_cpu_oracle():
    """min-marginals of a fresh solver (nothing deferred) equal the CPU oracle's, so the type counts can be derived without
    touching the GPU results at all."""
    col, costs = random_set_cover(800, 700, 6, seed=17)
    o = Oracle(col, costs, "double")
    omm = o.min_marginals()                   # BDD-major
    ovar, obdd = o.layer_info()
    order = np.lexsort((obdd, ovar))          # (variable, bdd)
    t = R.compute_mm_types(col.nr_variables(), ovar[order], omm[order, 0], omm[order, 1])
    s = bdd_hip_parallel_mma(col, costs, precision="double")
    res = s.perturb_primal_costs(0.1)
    assert res["counts"] == R.counts(t)
    c0, c1, side = R.perturbation(t, *R.compute_mm_sums(col.nr_variables(), ovar[order], omm[order, 0], omm[order, 1], np.float64), 0.1, np.float64)
    det = side == -2
    np.testing.assert_array_equal(res["cost_delta_0"][det], c0[det])
    np.testing.assert_array_equal(res["cost_delta_1"][det], c1[det])


def test_all_variables_agree_reads_off_the_solution():
    ilp = assignment_ilp(3)   # diagonal costs -2: the LP relaxation is tight, every min-marginal difference has a sign
    col = to_bdd_collection(ilp)
    s = bdd_hip_parallel_mma(col, ilp.objective, precision="double")
    s.iterations(50)
    lo, hi, _ = s.get_solver_costs()
    s.distribute_delta()
    lo, hi, _ = s.get_solver_costs()
    res = s.perturb_primal_costs(0.1)
    assert res["counts"][0] + res["counts"][1] == 9 and res["counts"][0] == 3
    assert float(np.dot(res["sol"], ilp.objective)) == -6.0
    lo2, hi2, _ = s.get_solver_costs()
    np.testing.assert_array_equal(lo2, lo); np.testing.assert_array_equal(hi2, hi)   # costs left alone (:295-305)


def test_rounding_with_lbfgs_flushes_the_history():
    """perturb_primal_costs calls s.update_costs on the L-BFGS type, which drops the (s, y) history first
    (lbfgs_impl.h:343-364); ADVICE r1: the rounding loop used to keep curvature pairs of the old objective."""
    col, costs = random_set_cover(1500, 1200, 6, seed=9)
    s = bdd_hip_parallel_mma(col, costs, precision="double")
    l = bdd_hip_lbfgs(s)
    for _ in range(12):
        l.iteration()
    assert l.state()["history_entries"] == 5
    res = s.perturb_primal_costs(0.2, round_index=0, seed=1, lbfgs=l)
    assert res["counts"][0] + res["counts"][1] < s.nr_variables()
    st = l.state()
    assert st["history_entries"] == 0 and st["num_unsuccessful_updates"] == 0
    # the history is rebuilt from scratch on the perturbed objective and the bound stays monotone (lbfgs_impl.h:403)
    prev = l.lower_bound()
    for _ in range(15):
        l.iteration()
        cur = l.lower_bound()
        assert cur >= prev - 1e-6
        prev = cur
    assert l.state()["history_entries"] == 5
    # the full rounding loop with an L-BFGS handle finds a feasible primal
    from bdd_amd import capi
    import ctypes as C
    sol = np.zeros(s.nr_variables(), np.int8)
    found = C.c_int(0)
    capi.check(capi.lib().bddmma_incremental_mm_agreement_rounding(s._h, l._h, 0.1, 1.2, 30, 200, 3, 0, sol.ctypes.data_as(C.c_void_p),
                                                                   C.byref(found)), s._h)
    assert found.value == 1
    x = sol.astype(float)
    assert all(col.evaluate(b, x) for b in range(col.nr_bdds()))
