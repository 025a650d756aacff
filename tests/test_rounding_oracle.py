"""Hand-computed cases for oracle/rounding_oracle.py (the numpy restatement of the reference's rounding functors,
incremental_mm_agreement_rounding_cuda.cu:29-205) and for oracle/lbfgs_oracle.py — CPU only."""
import numpy as np

from bdd_amd.instances import random_set_cover
from oracle import rounding_oracle as R
from oracle.lbfgs_oracle import LbfgsOracle
from oracle.oracle import Oracle


def test_direction_thresholds():
    # mm_diff_direction_func :29-41: -1 iff mm0 + 1e-6 <= mm1, +1 iff mm1 + 1e-6 <= mm0, else 0
    mm0 = np.array([0.0, 0.0, 0.0, 1.0, 1.0, 5.0])
    mm1 = np.array([1e-6, 0.9e-6, -1e-6, 1.0, 2.0, 4.0])
    np.testing.assert_array_equal(R.mm_diff_direction(mm0, mm1), [-1, 0, 1, 0, -1, 1])
    # float inputs are promoted to double before the 1e-6 is added (the literal is a double)
    a = np.float32(1.0)
    b = np.float32(1.0) + np.float32(1.1920929e-07) * 8   # 1 + 8 ulp = 1 + 9.5e-7 < 1 + 1e-6
    assert R.mm_diff_direction(np.array([a]), np.array([b]))[0] == 0


def test_type_table_and_counts():
    # variable 0: all layers prefer 1 -> one; 1: all prefer 0 -> zero; 2: all equal; 3: mixed signs; 4: one equal + one "one";
    # 5: in no BDD
    var = np.array([0, 0, 1, 1, 2, 2, 3, 3, 4, 4])
    mm0 = np.array([3.0, 2.0, 0.0, 1.0, 1.0, 2.0, 0.0, 5.0, 1.0, 4.0])
    mm1 = np.array([1.0, 1.0, 1.0, 3.0, 1.0, 2.0, 1.0, 1.0, 1.0, 2.0])
    t = R.compute_mm_types(6, var, mm0, mm1)
    # fill_mm_type_func :43-65: mm_min > 0 one; mm_max < 0 zero; both 0 equal; else inconsistent (4: min 0, max 1)
    np.testing.assert_array_equal(t, [R.ONE, R.ZERO, R.EQUAL, R.INCONSISTENT, R.INCONSISTENT, R.ZERO])
    assert R.counts(t) == (1, 2, 1, 2)
    s0, s1 = R.compute_mm_sums(6, var, mm0, mm1, np.float64)
    np.testing.assert_array_equal(s0, [5, 1, 3, 5, 5, 0])
    np.testing.assert_array_equal(s1, [2, 4, 3, 2, 3, 0])
    c0, c1, side = R.perturbation(t, s0, s1, 0.25, np.float64)
    # mm_types_transform :136-205: one -> {delta, 0}; zero -> {0, delta}
    np.testing.assert_array_equal(c0, [0.25, 0, 0, 0, 0, 0])
    np.testing.assert_array_equal(c1, [0, 0.25, 0, 0, 0, 0.25])
    # inconsistent: mm_0 < mm_1 -> {0, |r| delta} else {|r| delta, 0}; equal: sign of the draw
    np.testing.assert_array_equal(side, [-2, -2, -1, 0, 0, -2])


def test_lbfgs_oracle_state_machine():
    """The restated L-BFGS loop on the CPU oracle: collects `history_size` curvature pairs with plain MMA iterations, then
    takes L-BFGS steps; the bound never decreases (assert at lbfgs_impl.h:403) and beats plain MMA."""
    col, costs = random_set_cover(600, 500, 6, seed=4)
    l = LbfgsOracle(Oracle(col, costs, "double"))
    plain = Oracle(col, costs, "double")
    lbs, kinds = [], []
    for _ in range(40):
        l.iteration()
        plain.iteration()
        lbs.append(l.lower_bound())
        kinds.append(l.last_kind)
    assert kinds[:5] == [0] * 5                      # history.size() < m: mma iterations (choose_solver, :409-417)
    assert sum(kinds) >= 10                          # L-BFGS steps do happen
    assert all(b >= a - 1e-6 for a, b in zip(lbs, lbs[1:]))
    assert lbs[-1] >= plain.lower_bound() - 1e-9
    assert len(l.history) <= l.m
    # update_costs drops the history (:343-364)
    l.update_costs([], np.zeros(col.nr_variables()))
    assert len(l.history) == 0 and not l.prev_states_stored and l.num_unsuccessful == 0
