"""Pin the second oracle (oracle/cuda_rule_oracle.c: the reference's GPU-solver semantics with omega and the GPU's rule for
non-finite min-marginals) — CPU only.

Its finite branch at omega = 0.5 must reproduce, bit for bit, the golden traces that the reference's own compiled node arithmetic
produced (oracle/_ref via oracle/make_golden.py) and the pinned CPU oracle; the omega and non-finite branches are restated from
bdd_cuda_parallel_mma.cu:29-42 and are checked here by the properties the reference states for them."""
import numpy as np
import pytest

from bdd_amd import BddCollection, parse_lp, to_bdd_collection
from bdd_amd.instances import assignment_ilp, brute_force_optimum, random_set_cover
from oracle.oracle import CudaRuleOracle, Oracle
from test_oracle_kat import SIMPLEX_KATS
from util import GOLDEN, load_golden, pad_costs, suffix


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("precision", ["double", "float"])
def test_finite_branch_matches_reference_traces_bit_for_bit(name, precision):
    col, z = load_golden(name)
    sfx = suffix(precision)
    dt = np.float64 if precision == "double" else np.float32
    o = CudaRuleOracle(col, None, precision)
    V = o.nr_variables()
    o.update_costs([], pad_costs(z["costs"], V))
    assert o.lower_bound() == float(z[f"lb_init_{sfx}"])
    d = np.zeros(2 * V, dt)
    for it in range(10):
        o.forward_mm(0.5, d)
        np.testing.assert_array_equal(d, z[f"delta_trace_{sfx}"][it, 0])
        o.backward_mm(0.5, d)
        np.testing.assert_array_equal(d, z[f"delta_trace_{sfx}"][it, 1])
        assert o.lower_bound() == float(z[f"lb_trace_{sfx}"][it])
    o2 = CudaRuleOracle(col, None, precision)
    o2.update_costs([], pad_costs(z["costs"], V))
    for it in range(20):
        o2.iteration(0.5)
        assert o2.lower_bound() == float(z[f"iter_lb_{sfx}"][it])


@pytest.mark.parametrize("precision", ["double", "float"])
def test_agrees_with_the_cpu_oracle_per_layer(precision):
    """costs, deferred differences and min-marginals per layer, on a seeded cover instance (all min-marginals finite)"""
    col, costs = random_set_cover(400, 300, 7, seed=11)
    o = Oracle(col, costs, precision)
    c = CudaRuleOracle(col, costs, precision)
    np.testing.assert_array_equal(np.stack(c.layer_info()), np.stack(o.layer_info()))
    for it in range(8):
        o.iteration(); c.iteration(0.5)
        assert c.lower_bound() == o.lower_bound()
        np.testing.assert_array_equal(c.delta(), o.delta_in())
        np.testing.assert_array_equal(c.mm(), o.mm_last())
        for a, b in zip(c.get_costs(), o.get_costs()):
            np.testing.assert_array_equal(a, b)
    mm0, mm1 = c.min_marginals()
    omm = o.min_marginals()
    tol = dict(rtol=1e-12, atol=1e-12) if precision == "double" else dict(rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(mm0, omm[:, 0], **tol)   # path costs associate as F + (T + c) on the GPU, (F + c) + T on the CPU
    np.testing.assert_allclose(mm1, omm[:, 1], **tol)


@pytest.mark.parametrize("lp,nv,nb,lb", SIMPLEX_KATS + [(assignment_ilp(3).write_lp(), 9, 6, -6.0)])
def test_reference_known_answers(lp, nv, nb, lb):
    # test/test_bdd_cuda_base.cpp:49-115
    ilp = parse_lp(lp)
    for prec in ("float", "double"):
        o = CudaRuleOracle(to_bdd_collection(ilp), ilp.objective, prec)
        assert o.nr_variables() == nv and o.nr_bdds() == nb
        assert o.lower_bound() == lb


def _forced_ilp():
    from bdd_amd.ilp import ILP
    ilp = ILP()
    names = [f"x{i}" for i in range(10)]
    for n in names:
        ilp.var(n)
    ilp.objective = [3.0, -1.5, 2.0, 0.5, -2.5, 1.0, -0.5, 4.0, -3.0, 0.25]
    ilp.add_constraint([(1, "x3")], "=", 1)
    ilp.add_constraint([(1, "x7")], ">=", 1)
    ilp.add_constraint([(1, "x0"), (1, "x1")], "=", 2)
    ilp.add_constraint([(1, "x4"), (1, "x5")], "<=", 0)
    ilp.add_constraint([(1, n) for n in ("x0", "x2", "x4", "x6", "x8")], "<=", 3)
    ilp.add_constraint([(1, n) for n in ("x1", "x3", "x5", "x7", "x9")], ">=", 3)
    ilp.add_constraint([(2, "x2"), (-1, "x6"), (1, "x8"), (1, "x9")], ">=", 1)
    return ilp


@pytest.mark.parametrize("omega", [0.25, 0.5, 0.8, 1.0])
def test_non_finite_rule_and_omega_properties(omega):
    """Forced variables: a layer whose lo or hi side only reaches the bot sink keeps mm = 0 and its costs only take the incoming
    delta (bdd_cuda_parallel_mma.cu:36-39); all costs stay finite (the asserts of :203-204,293-294); the bound is monotone and
    never exceeds the optimum; the sum of all arc costs + pending deltas is what the reparametrisation preserves."""
    ilp = _forced_ilp()
    opt = brute_force_optimum(ilp)
    col = to_bdd_collection(ilp)
    o = CudaRuleOracle(col, ilp.objective, "double")
    var, _ = o.layer_info()
    prev = o.lower_bound()
    assert np.isfinite(prev) and prev <= opt + 1e-9
    seen_zero = False
    for _ in range(60):
        lo0, hi0 = o.get_costs()
        o.iteration(omega)
        lo, hi = o.get_costs()
        assert np.all(np.isfinite(lo)) and np.all(np.isfinite(hi))
        mm0, mm1 = o.min_marginals()
        forced = ~(np.isfinite(mm0) & np.isfinite(mm1))
        assert forced.any()
        seen_zero |= bool(np.all(o.mm()[forced] == 0))
        lb = o.lower_bound()
        assert lb >= prev - 1e-9 and lb <= opt + 1e-9
        prev = lb
    assert seen_zero


def test_omega_scales_the_first_pass_linearly():
    """One forward pass from equal states: the deferred differences are omega * (m1 - m0), i.e. mm(omega) / omega is the same
    for every omega (exactly, for powers of two)."""
    col, costs = random_set_cover(200, 150, 6, seed=3)
    ref = None
    for omega in (0.25, 0.5, 1.0):
        o = CudaRuleOracle(col, costs, "double")
        d = np.zeros(2 * o.nr_variables())
        o.forward_mm(omega, d)
        # only the first layer of every BDD has seen no omega-dependent cost yet
        first = np.r_[True, np.diff(o.layer_info()[1]) != 0]
        m = o.mm()[first] / omega
        if ref is None:
            ref = m
        np.testing.assert_array_equal(m, ref)
