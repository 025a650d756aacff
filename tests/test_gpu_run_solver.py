"""run_solver (include/run_solver_util.h:10-77) with its termination tests on the device (solver_impl.hpp: run_plain,
kernels.hpp: k_lb_reduce_ctl) against the reference's sequential loop — iteration(); lower_bound(); tests — restated here
in Python on a twin solver, and against the library's own sequential loop (bddmma_run_solver_host_loop)."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from bdd_amd.instances import random_set_cover  # noqa: E402
from bdd_amd.solver import bdd_hip_parallel_mma, run_solver  # noqa: E402


def sequential_loop(s, max_iter, tolerance, slope):
    """run_solver_util.h:27-73 without the wall-clock test; returns (iterations, lb_final, reason, lb_initial)."""
    lb_initial = s.lower_bound()
    lb_first = float(np.finfo(np.float64).max)
    lb_prev = lb_post = lb_initial
    reason = 0
    for it in range(max_iter):
        s.iteration()
        lb_prev, lb_post = lb_post, s.lower_bound()
        if it == 0:
            lb_first = lb_post
        if abs(lb_prev - lb_post) < abs(tolerance * lb_prev):
            reason = 2
        elif abs(lb_prev - lb_post) < slope * abs(lb_initial - lb_first):
            reason = 3
        elif lb_post == math.inf:
            reason = 4
        if reason:
            return it + 1, lb_post, reason, lb_initial
    return max_iter, lb_post, 0, lb_initial


CASES = [
    dict(max_iter=23, tolerance=0.0, slope=0.0),      # runs to the iteration limit
    dict(max_iter=400, tolerance=2e-4, slope=0.0),    # relative progress below the tolerance
    dict(max_iter=400, tolerance=0.0, slope=0.05),    # progress below 5 % of the first iteration's
    dict(max_iter=400, tolerance=1e-6, slope=1e-9),   # the JSON driver's defaults: whichever comes first
]


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("case", CASES, ids=["max_iter", "tolerance", "slope", "defaults"])
def test_device_resident_loop_equals_the_sequential_loop(precision, case):
    col, costs = random_set_cover(3000, 2500, 8, seed=21)
    # deterministic exchange: two solvers then produce the same bits, so iteration counts and bounds can be compared exactly
    twin = bdd_hip_parallel_mma(col, costs, precision=precision, deterministic=True)
    its, lb, reason, lb_initial = sequential_loop(twin, case["max_iter"], case["tolerance"], case["slope"])
    s = bdd_hip_parallel_mma(col, costs, precision=precision, deterministic=True)
    res = run_solver(s, max_iter=case["max_iter"], tolerance=case["tolerance"], improvement_slope=case["slope"], time_limit=1e9)
    assert (res["iterations"], res["stop_reason"]) == (its, reason)
    assert res["lb_final"] == lb and res["lb_initial"] == lb_initial
    if reason:
        assert its < case["max_iter"]  # the case does exercise its criterion
    # the iterations queued behind the stopping one were not executed: same state as the twin, bit for bit
    assert s.lower_bound() == lb
    for a, b in zip(s.get_solver_costs(), twin.get_solver_costs()):
        np.testing.assert_array_equal(a, b)
    # ... and the solver carries on from there like the twin
    s.iteration(); twin.iteration()
    assert s.lower_bound() == twin.lower_bound()


def test_default_exchange_and_the_librarys_own_sequential_loop():
    col, costs = random_set_cover(4000, 3000, 10, seed=5)
    results = []
    for host_loop in (False, True):
        s = bdd_hip_parallel_mma(col, costs, precision="double")
        results.append(run_solver(s, max_iter=300, tolerance=1e-5, improvement_slope=0.0, time_limit=1e9, host_loop=host_loop))
    a, b = results
    assert a["stop_reason"] == b["stop_reason"] == 2 and a["iterations"] == b["iterations"] < 300
    assert abs(a["lb_final"] - b["lb_final"]) <= 1e-12 * abs(b["lb_final"])  # LDS-atomic order only


def test_time_limit_and_degenerate_limits():
    col, costs = random_set_cover(2000, 1500, 8, seed=2)
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    lb0 = s.lower_bound()
    res = run_solver(s, max_iter=0, tolerance=1e-6, improvement_slope=0.0, time_limit=10)
    assert res["iterations"] == 0 and res["lb_final"] == lb0 == res["lb_initial"] and res["stop_reason"] == 0
    # time limit 0: the reference runs one iteration, then "Time limit reached" (run_solver_util.h:50-55)
    res = run_solver(s, max_iter=100, tolerance=0.0, improvement_slope=0.0, time_limit=0.0)
    assert res["iterations"] == 1 and res["stop_reason"] == 1 and res["lb_final"] >= lb0
    assert s.lower_bound() == res["lb_final"]
    # a limit in the middle of a long run: stops by the clock, bound of the last iteration that ran
    res = run_solver(s, max_iter=10**7, tolerance=0.0, improvement_slope=0.0, time_limit=0.05)
    # (upper bound: generous — with eight processes time-slicing the GPU and oversubscribing the shared host in tools/soak.sh a queue, or the
    #  host thread itself, can stay descheduled for a long time; the clock keeps running, the run stops at its next test and reports the real
    #  elapsed time: 4.07 s once in 32 loaded runs in round 4, 52.9 s once in round 6 — with 377 iterations and the right reason)
    assert res["stop_reason"] == 1 and 1 <= res["iterations"] < 10**7 and 0.05 <= res["seconds"] < 600.0
    assert s.lower_bound() == res["lb_final"]
    # the clock is tested on the device with the other criteria (run_ctl_step): the iterations queued behind the one that crossed the limit
    # did not run — the state is the one after exactly `iterations` iterations (the host-side test let up to five more execute)
    a = bdd_hip_parallel_mma(col, costs, precision="double", deterministic=True)
    b = bdd_hip_parallel_mma(col, costs, precision="double", deterministic=True)
    res = run_solver(a, max_iter=10**7, tolerance=0.0, improvement_slope=0.0, time_limit=0.03)
    assert res["stop_reason"] == 1 and res["iterations"] > 10
    b.iterations(res["iterations"])
    assert a.lower_bound() == b.lower_bound() == res["lb_final"]
    for x, y in zip(a.get_solver_costs(), b.get_solver_costs()):
        np.testing.assert_array_equal(x, y)
    # and the solver is reusable afterwards (the stop flag of one run does not leak into the next calls)
    before = s.lower_bound()
    s.iterations(3)
    assert s.lower_bound() >= before
    res = run_solver(s, max_iter=5, tolerance=0.0, improvement_slope=0.0, time_limit=1e9)
    assert res["iterations"] == 5 and res["stop_reason"] == 0


def test_wide_and_mixed_pack_kinds_stop_too():
    """every kernel family of an iteration honours the stop flag: narrow + wide packs in one instance"""
    from bdd_amd import BddCollection
    rng = np.random.Generator(np.random.PCG64(3))
    col = BddCollection()
    V = 400
    for _ in range(120):
        vs = np.sort(rng.choice(V, size=12, replace=False))
        co = rng.integers(1, 30, size=12)
        col.add_linear(co, "<=", int(co.sum() // 2), vs)
    rows = np.sort(np.stack([rng.choice(V, size=6, replace=False) for _ in range(300)]), axis=1).astype(np.uint64)
    col.add_covering(rows)
    costs = rng.uniform(-5, 5, col.nr_variables())
    twin = bdd_hip_parallel_mma(col, costs, precision="double", deterministic=True)
    its, lb, reason, _ = sequential_loop(twin, 500, 1e-4, 0.0)
    s = bdd_hip_parallel_mma(col, costs, precision="double", deterministic=True)
    res = run_solver(s, max_iter=500, tolerance=1e-4, improvement_slope=0.0, time_limit=1e9)
    assert (res["iterations"], res["stop_reason"], res["lb_final"]) == (its, reason, lb) and reason == 2
    for a, b in zip(s.get_solver_costs(), twin.get_solver_costs()):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("mode", ["deterministic", "by_variable", "small_bins"])
def test_stopping_launch_completes_its_own_grid(mode):
    """the launch that latches the stop word must not skip part of its own grid.  With thousands of exchange
    workgroups (6 836 k_exchange_bcast / 5 469 k_exchange_byvar blocks, 21 875 bins of 64 variables) most of them are dispatched
    after workgroup 0 has run the termination tests; a plain flag made those return without writing their delta pairs.  After
    run_solver the deferred delta, the costs and one further iteration must equal the sequential twin's."""
    from bdd_amd.instances import random_set_cover_mt
    col, costs = random_set_cover_mt(1_400_000, 700_000, 10, 77)
    opts = {"deterministic": dict(deterministic=True), "by_variable": dict(exchange_by_variable=2), "small_bins": dict(vars_per_bin=64)}[mode]
    exact = mode != "small_bins"   # LDS-atomic accumulation order differs between two runs of the default exchange
    twin = bdd_hip_parallel_mma(col, costs, precision="double", **opts)
    its, lb, reason, _ = sequential_loop(twin, 200, 3e-3, 0.0)
    assert reason == 2 and 3 <= its < 200
    s = bdd_hip_parallel_mma(col, costs, precision="double", **opts)
    # The race needs late dispatch, i.e. a busy GPU (the round-2 library passed this test on an idle one and failed a third of the
    # run_solver test runs with four processes on the GPU, profiles/r03_soak_*.txt): a second solver iterates on its own stream
    # from a second host thread while run_solver runs (ctypes releases the GIL during the calls).
    import threading
    other = bdd_hip_parallel_mma(col, costs, precision="float")
    busy = threading.Event()

    def load():
        while not busy.is_set():
            other.iterations(20)
    t = threading.Thread(target=load)
    t.start()
    try:
        res = run_solver(s, max_iter=200, tolerance=3e-3, improvement_slope=0.0, time_limit=1e9)
    finally:
        busy.set()
        t.join()
    assert (res["iterations"], res["stop_reason"]) == (its, reason)

    def same(a, b):
        if exact:
            np.testing.assert_array_equal(a, b)
        else:
            np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-11)
    assert res["lb_final"] == lb if exact else abs(res["lb_final"] - lb) <= 1e-11 * abs(lb)
    same(s.get_delta(), twin.get_delta())
    for a, b in zip(s.get_solver_costs(), twin.get_solver_costs()):
        same(a, b)
    s.iteration(); twin.iteration()
    a, b = s.lower_bound(), twin.lower_bound()
    assert a == b if exact else abs(a - b) <= 1e-11 * abs(b)
    same(s.get_delta(), twin.get_delta())
