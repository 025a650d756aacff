"""HIP path vs oracle/cuda_rule_oracle.c — the two branches the CPU `parallel mma` oracle cannot pin:

  * omega != 0.5 (bdd_cuda_parallel_mma.cu:29-42,142-153: the GPU solver scales by the omega it is given; the CPU solver hard-codes 0.5,
    bdd_parallel_mma_base.cpp:975,1001);
  * non-finite min-marginals (`mm = 0` unless both are finite, bdd_cuda_parallel_mma.cu:36-39, where the CPU solver writes +inf costs).

Everything goes through the C-ABI.  Double runs are held to 1e-9, float runs to 1e-5 relative (BASELINE.json's tolerance); values that
must be exact (a forced layer's deferred difference, infinities of the min-marginals) are compared exactly.  Needs an MI355X.
"""
import os

import numpy as np
import pytest

from bdd_amd import BddCollection, to_bdd_collection
from bdd_amd.instances import brute_force_optimum
from bdd_amd.solver import bdd_hip_parallel_mma
from oracle.oracle import CudaRuleOracle
from test_cuda_rule_oracle import _forced_ilp
from util import GOLDEN, load_golden, pad_costs

pytestmark = pytest.mark.gpu

TOL = {"double": dict(abs=1e-9, rel=1e-9), "float": dict(abs=2e-4, rel=1e-5)}
OMEGAS = [0.25, 0.8, 1.0]


def layer_perm(s, o):
    """internal layer order -> oracle (BDD-major) order"""
    perm = s.bdd_major_order()
    var, bdd = o.layer_info()
    np.testing.assert_array_equal(s.get_primal_variable_index()[perm], var)
    np.testing.assert_array_equal(s.get_bdd_index()[perm], bdd)
    return perm


def check_state(s, o, precision, scale, what=""):
    """lower bound, arc costs and deferred differences of the solver against the oracle's"""
    t = TOL[precision]
    lb, ref = s.lower_bound(), o.lower_bound()
    assert np.isfinite(ref) and abs(lb - ref) <= t["abs"] * scale + t["rel"] * abs(ref), (what, lb, ref)
    perm = layer_perm(s, o)
    lo, hi, mm = s.get_solver_costs()
    olo, ohi = o.get_costs()
    kw = dict(atol=t["abs"] * scale, rtol=t["rel"] * 10, err_msg=what)
    np.testing.assert_allclose(lo[perm], olo, **kw)
    np.testing.assert_allclose(hi[perm], ohi, **kw)
    np.testing.assert_allclose(mm[perm], o.mm(), **kw)
    return perm


def protocol(s, o, omega, precision, scale, n_it, what=""):
    """test/test_cuda_parallel_mma.cu:60-99 with omega as given: delta after every pass, bound after every iteration"""
    t = TOL[precision]
    V = s.nr_variables()
    d = np.zeros(2 * V, s.value_type)
    r = np.zeros(2 * V, s.value_type)
    for it in range(n_it):
        s.forward_mm(omega, d); o.forward_mm(omega, r)
        np.testing.assert_allclose(d, r, atol=t["abs"] * scale, rtol=t["rel"] * 10, err_msg=f"{what} forward {it}")
        s.normalize_delta(d); o.normalize_delta(r)
        s.backward_mm(omega, d); o.backward_mm(omega, r)
        np.testing.assert_allclose(d, r, atol=t["abs"] * scale, rtol=t["rel"] * 10, err_msg=f"{what} backward {it}")
        s.normalize_delta(d); o.normalize_delta(r)
        lb, ref = s.lower_bound(), o.lower_bound()
        assert abs(lb - ref) <= t["abs"] * scale + t["rel"] * abs(ref), (what, it, lb, ref)


# ---------------------------------------------------------------- omega
@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("omega", OMEGAS)
def test_omega_protocol_on_golden_instances(name, precision, omega):
    col, z = load_golden(name)
    costs = pad_costs(z["costs"], col.nr_variables())
    scale = max(1.0, float(np.abs(z["costs"]).max()))
    for pw in (64, 128):
        s = bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=pw)
        o = CudaRuleOracle(col, costs, precision)
        protocol(s, o, omega, precision, scale, 8, f"{name} pw={pw}")
        check_state(s, o, precision, scale, name)
    # iteration(omega): the solver's own delta_lo_hi_
    s = bdd_hip_parallel_mma(col, costs, precision=precision)
    o = CudaRuleOracle(col, costs, precision)
    for it in range(12):
        s.iteration(omega); o.iteration(omega)
        lb, ref = s.lower_bound(), o.lower_bound()
        assert abs(lb - ref) <= TOL[precision]["abs"] * scale + TOL[precision]["rel"] * abs(ref), (it, lb, ref)
    check_state(s, o, precision, scale, name)


def wide_staggered_instance(seed=77, n_wide=14, n_cover=40, V=60):
    rng = np.random.Generator(np.random.PCG64(seed))
    col = BddCollection()
    for _ in range(n_wide):
        k = int(rng.integers(15, 20))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        co = rng.integers(1, 40, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), vs)
    for _ in range(n_cover):
        col.add_covering(np.sort(rng.choice(V, size=5, replace=False)))
    costs = rng.normal(0, 3, col.nr_variables()).round(3)
    widest = max(np.bincount(col.instr[int(col.delims[b]):int(col.delims[b + 1]) - 2, 2].astype(np.int64)).max() for b in range(n_wide))
    return col, costs, int(-(-widest // 64) * 64)


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("omega", OMEGAS)
@pytest.mark.parametrize("kind", ["wide", "staggered", "separate-launches", "huge"])
def test_omega_on_wide_staggered_and_huge_packs(precision, omega, kind):
    col, costs, wpw = wide_staggered_instance()
    opts = {"wide": dict(pack_width=64, wide_pack_width=512),
            "staggered": dict(pack_width=64, wide_pack_width=wpw, pack_stagger=60),
            "separate-launches": dict(pack_width=64, wide_pack_width=wpw, pack_stagger=60, variant_flags=3),
            "huge": dict(pack_width=64, wide_pack_width=64)}[kind]   # frontier in global memory (k_*_wide<GLOBAL>)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, **opts)
    o = CudaRuleOracle(col, costs, precision)
    protocol(s, o, omega, precision, 100.0, 6, kind)
    for _ in range(6):
        s.iteration(omega); o.iteration(omega)
    check_state(s, o, precision, 100.0, kind)


def fuzz_rows(rng, V, forcing):
    """rows of the differential fuzz (test_gpu_parity.py); forcing = True adds rows that fix variables: single-variable rows, rows
    whose right-hand side leaves one value to some variables, coefficients larger than the right-hand side"""
    rows = []
    for _ in range(int(rng.integers(20, 200))):
        kind = rng.integers(0, 5)
        k = int(rng.integers(2, min(V, 14) + 1))
        vs = rng.choice(V, size=k, replace=False)
        if rng.random() < 0.7:
            vs = np.sort(vs)
        if kind == 0:
            rows.append((np.ones(k, int), vs, ">=", 1))
        elif kind == 1:
            rows.append((np.ones(k, int), vs, "=", 1))
        elif kind == 2 and k >= 3:
            rows.append((np.ones(k, int), vs, "=", int(rng.integers(1, k))))
        elif kind == 3:
            co = rng.integers(1, 9, size=k)
            rows.append((co, vs, "<=", int(rng.integers(co.max(), co.sum()))))
        else:
            co = rng.integers(1, 9, size=k)
            rows.append((co, vs, ">=", int(rng.integers(1, co.sum() - co.max() + 1))))
    if forcing:
        # the instance stays feasible: x[v] = assign[v] satisfies every forcing row by construction
        assign = rng.integers(0, 2, size=V)
        for _ in range(int(rng.integers(4, 30))):
            kind = rng.integers(0, 5)
            k = int(rng.integers(1, min(V, 12) + 1))
            vs = np.sort(rng.choice(V, size=k, replace=False))
            a = assign[vs]
            if kind == 0:                                     # single variable, fixed to its value
                v = vs[:1]
                rows.append((np.ones(1, int), v, "=", int(assign[v[0]])))
            elif kind == 1:                                   # sum x = number of ones: fixes every variable of the row
                rows.append((np.ones(k, int), vs, "=", int(a.sum())))
            elif kind == 2:                                   # knapsack in which the variables at 0 do not fit
                co = rng.integers(1, 9, size=k)
                rhs = int((co * a).sum())
                co = np.where(a == 0, rhs + rng.integers(1, 5, size=k), co)
                rows.append((co, vs, "<=", rhs))
            elif kind == 3:                                   # covering with weights in which the variables at 1 are all needed
                co = rng.integers(1, 9, size=k)
                rows.append((co, vs, ">=", int((co * a).sum()) if a.any() else 0))
            else:                                             # wide layers (coefficients up to 60) with some variables that do not fit
                k = int(rng.integers(10, min(V, 18) + 1))
                vs = np.sort(rng.choice(V, size=k, replace=False))
                a = assign[vs]
                co = rng.integers(1, 60, size=k)
                rhs = int((co * a).sum() + rng.integers(0, 30))
                co = np.where((a == 0) & (rng.random(k) < 0.3), rhs + 1 + rng.integers(0, 9, size=k), co)
                rows.append((co, vs, "<=", rhs))
        rows = [r for r in rows if not (r[2] == ">=" and r[3] <= 0)]
        # the earlier free rows must hold for `assign` too, or the instance may be infeasible (bound +inf on both sides: nothing to compare)
        keep = []
        for co, vs, op, rhs in rows:
            val = int((np.asarray(co) * assign[np.asarray(vs)]).sum())
            if (op == ">=" and val >= rhs) or (op == "<=" and val <= rhs) or (op == "=" and val == rhs):
                keep.append((co, vs, op, rhs))
        rows = keep
    return rows


@pytest.mark.parametrize("seed", range(int(os.environ.get("BDDMMA_FUZZ_SEEDS", "16"))))
def test_omega_and_forced_variables_fuzz(seed):
    """Random instances under random layout options (every pack kind, resident / streaming sweeps, both entry orders), random omega,
    odd seeds with rows that force variables: bound, arc costs, deferred differences and min-marginals against the GPU-rule oracle."""
    from bdd_amd import native
    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    V = int(rng.integers(30, 300))
    forcing = seed % 2 == 1
    rows = fuzz_rows(rng, V, forcing)
    if seed % 4 >= 2:
        for _ in range(int(rng.integers(1, 6))):
            k = int(rng.integers(12, min(V, 20) + 1))
            co = rng.integers(1, 60, size=k)
            rows.append((co, np.sort(rng.choice(V, size=k, replace=False)), "<=", int(co.sum())))   # wide and always satisfied
    col = native.rows_to_bdd_collection(rows)
    if col.nr_bdds() == 0:
        pytest.skip("all rows trivial")
    costs = rng.normal(0, 4, col.nr_variables()).round(3)
    omega = float(rng.choice([0.25, 0.5, 0.6, 0.8, 1.0]))
    pw = int(rng.choice([64, 128, 256]))
    wpb = int(rng.choice([1, 2, 4, 8]))
    cap = int(rng.choice([pw, 256, 640])) if wpb < 8 else 256
    opts = dict(pack_width=pw, waves_per_block=wpb, stage_cap=max(cap, pw), vars_per_bin=int(rng.choice([0, 64, 256])),
                wide_pack_width=int(rng.choice([0, 64, 128, 256])), keep_bdd_order=bool(rng.integers(0, 2)),
                resident_sweeps=int(rng.choice([0, 1, 2])), exchange_by_variable=int(rng.choice([0, 0, 2])),
                variant_flags=int(rng.choice([0, 0, 1, 2, 3])) | int(rng.choice([0, 0x800, 0x1000, 0x2000, 0x2000])) | int(rng.choice([0, 0, 0x4000])), pack_fill=int(rng.choice([0, 0, pw // 2, 16])),
                pack_stagger=int(rng.choice([0, 1, 24, 60, 200])))
    what = f"seed {seed} omega {omega} {opts}"
    n_it = int(rng.integers(3, 20))
    for precision in ("double", "float"):
        s = bdd_hip_parallel_mma(col, costs, precision=precision, **opts)
        o = CudaRuleOracle(col, costs, precision)
        if not np.isfinite(o.lower_bound()):
            pytest.skip("infeasible row mixture")
        protocol(s, o, omega, precision, 100.0, 2, what)
        for _ in range(n_it):
            s.iteration(omega); o.iteration(omega)
        perm = check_state(s, o, precision, 100.0, what)
        _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
        r0, r1 = o.min_marginals()
        if forcing:
            assert not (np.isfinite(r0) & np.isfinite(r1)).all(), "the forcing rows force nothing"
        np.testing.assert_array_equal(np.isfinite(mm0[perm]), np.isfinite(r0), err_msg=what)
        np.testing.assert_array_equal(np.isfinite(mm1[perm]), np.isfinite(r1), err_msg=what)
        t = TOL[precision]
        f0, f1 = np.isfinite(r0), np.isfinite(r1)
        np.testing.assert_allclose(mm0[perm][f0], r0[f0], atol=t["abs"] * 100, rtol=t["rel"] * 10, err_msg=what)
        np.testing.assert_allclose(mm1[perm][f1], r1[f1], atol=t["abs"] * 100, rtol=t["rel"] * 10, err_msg=what)


# ---------------------------------------------------------------- non-finite min-marginals
FORCED_OPTS = [dict(), dict(pack_width=64, waves_per_block=1), dict(pack_width=256, waves_per_block=8, stage_cap=256),
               dict(resident_sweeps=1), dict(resident_sweeps=2), dict(wide_pack_width=64, pack_width=64), dict(exchange_by_variable=2),
               dict(deterministic=1), dict(pack_stagger=24, keep_bdd_order=True), dict(variant_flags=0x2000, resident_sweeps=1),
               dict(variant_flags=0x2000, resident_sweeps=1, pack_stagger=24, keep_bdd_order=True), dict(variant_flags=0x1800),
               dict(variant_flags=0x4000), dict(variant_flags=0x6000)]


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("opts", FORCED_OPTS, ids=[",".join(f"{k}={v}" for k, v in o.items()) or "default" for o in FORCED_OPTS])
def test_forced_variables_vs_gpu_rule_oracle(precision, opts):
    """The instance of test_forced_variables_keep_the_bound_finite_and_valid (single-variable BDDs, rows that fix variables, bot-only
    arcs), now value for value: delta per pass, bound per iteration, arc costs, deferred differences (exactly 0 on forced layers)."""
    ilp = _forced_ilp()
    opt = brute_force_optimum(ilp)
    col = to_bdd_collection(ilp)
    for omega in (0.5, 0.8):
        s = bdd_hip_parallel_mma(col, ilp.objective, precision=precision, **opts)
        o = CudaRuleOracle(col, ilp.objective, precision)
        protocol(s, o, omega, precision, 10.0, 25, f"forced {opts}")
        perm = check_state(s, o, precision, 10.0, f"forced {opts}")
        assert s.lower_bound() <= opt + 1e-4
        _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
        r0, r1 = o.min_marginals()
        forced = ~(np.isfinite(r0) & np.isfinite(r1))
        assert forced.sum() >= 6
        np.testing.assert_array_equal(np.isinf(mm0[perm]), np.isinf(r0))
        np.testing.assert_array_equal(np.isinf(mm1[perm]), np.isinf(r1))
        s.iteration(omega); o.iteration(omega)
        _, _, mm = s.get_solver_costs()
        assert np.all(mm[perm][forced] == 0) and np.all(o.mm()[forced] == 0)       # no update, exactly
        assert np.any(mm[perm][~forced] != 0)


@pytest.mark.parametrize("precision", ["double", "float"])
def test_forced_variables_in_wide_and_huge_packs(precision):
    """Wide rows (layers > 64 nodes) in which some variables do not fit, next to single-variable rows on the same variables: the
    non-finite rule inside k_*_wide2, k_*_mixed and the global-memory frontier kernels."""
    from bdd_amd import native
    rng = np.random.Generator(np.random.PCG64(404))
    V = 40
    assign = rng.integers(0, 2, size=V)
    rows = []
    for _ in range(8):
        k = int(rng.integers(18, 24))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        a = assign[vs]
        co = rng.integers(1, 150, size=k)
        rhs = int((co * a).sum() + rng.integers(20, 120))
        co = np.where((a == 0) & (rng.random(k) < 0.15), rhs + 1 + rng.integers(0, 9, size=k), co)
        rows.append((co, vs, "<=", rhs))
    assert sum(max(w) > 64 for w in (native.rows_to_bdd_collection(rows).layer_widths(b) for b in range(8))) >= 4
    for v in rng.choice(V, size=6, replace=False):
        rows.append((np.ones(1, int), np.array([v]), "=", int(assign[v])))
    for _ in range(30):
        vs = np.sort(rng.choice(V, size=5, replace=False))
        if assign[vs].any():
            rows.append((np.ones(5, int), vs, ">=", 1))
    col = native.rows_to_bdd_collection(rows)
    costs = rng.normal(0, 3, col.nr_variables()).round(3)
    for opts in (dict(pack_width=64, wide_pack_width=512), dict(pack_width=64, wide_pack_width=0, pack_stagger=60),
                 dict(pack_width=64, wide_pack_width=64), dict(pack_width=64, wide_pack_width=256, variant_flags=3)):
        s = bdd_hip_parallel_mma(col, costs, precision=precision, **opts)
        o = CudaRuleOracle(col, costs, precision)
        assert np.isfinite(o.lower_bound())
        protocol(s, o, 0.5, precision, 100.0, 10, str(opts))
        perm = check_state(s, o, precision, 100.0, str(opts))
        r0, r1 = o.min_marginals()
        forced = ~(np.isfinite(r0) & np.isfinite(r1))
        assert forced.sum() >= 10
        _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
        np.testing.assert_array_equal(np.isinf(mm0[perm]), np.isinf(r0))
        np.testing.assert_array_equal(np.isinf(mm1[perm]), np.isinf(r1))
