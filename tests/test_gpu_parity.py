"""HIP path vs the CPU oracle / reference golden traces / reference known answers — needs an MI355X.

Everything goes through the C-ABI (bdd_amd.solver -> ctypes -> libbdd_mma_hip.so).
Tolerances: the reference's own CPU<->GPU test uses 1e-6 absolute on every delta and lower bound in
double (test/test_cuda_parallel_mma.cu:72-99); BASELINE.json asks for 1e-5 relative on the lower
bound after equal iterations.  Double runs here are held to 1e-9, float runs to 1e-5 relative.
"""
import os
import tempfile

import numpy as np
import pytest

from bdd_amd import BddCollection, parse_lp, to_bdd_collection
from bdd_amd.instances import GRID_3X3, LONG_CHAIN, SHORT_CHAIN, assignment_ilp, brute_force_optimum, mrf_ilp, random_set_cover, random_set_cover_mt
from bdd_amd.solver import bdd_hip_lbfgs, bdd_hip_parallel_mma, run_solver
from oracle.oracle import Oracle
from test_oracle_kat import SIMPLEX_KATS
from util import FULLSIZE, GOLDEN, load_golden, pad_costs, suffix, Checksum as _Checksum, parse_checkpoint as _parse_checkpoint, write_checkpoint as _write_checkpoint, CHECKPOINT_ARRAY_IDS

pytestmark = pytest.mark.gpu

TOL = {"double": dict(abs=1e-9, rel=1e-9), "float": dict(abs=2e-4, rel=1e-5)}


def close(a, b, precision, scale=1.0):
    t = TOL[precision]
    return abs(a - b) <= t["abs"] * max(1.0, scale) + t["rel"] * abs(b)


# ---------------------------------------------------------------- reference known answers
@pytest.mark.parametrize("lp,nv,nb,lb", SIMPLEX_KATS + [(assignment_ilp(3).write_lp(), 9, 6, -6.0)])
def test_bdd_cuda_base_kats(lp, nv, nb, lb):
    # test/test_bdd_cuda_base.cpp:49-115: bdd_cuda_base<float>, set_cost per variable, exact lower bound
    ilp = parse_lp(lp)
    s = bdd_hip_parallel_mma(to_bdd_collection(ilp), precision="float")
    assert s.nr_variables() == nv and s.nr_bdds() == nb
    for i, c in enumerate(ilp.objective):
        s.set_cost(c, i)
    assert s.lower_bound() == lb


def test_min_marginals_two_simplex():
    # test/test_bdd_cuda_min_marginals.cpp:17-36
    ilp = parse_lp(SIMPLEX_KATS[1][0])
    s = bdd_hip_parallel_mma(to_bdd_collection(ilp), precision="float")
    s.update_costs([], ilp.objective)
    mms = s.min_marginals()
    assert len(mms) == 6 and all(m.shape == (1, 2) for m in mms)
    expect = [(1, 2), (1, 1), (1, 1), (1, 0), (0, 1), (3, 0)]
    for m, e in zip(mms, expect):
        assert tuple(m[0]) == e


TWO_SIMPLEX_DIFF_SIZE = "Minimize\n2 x_1 + 1 x_2 + 1.5 x_3\n+2 x_4 + 2 x_5 + 3 x_6\nSubject To\nx_1 + x_2 + x_3 + x_4 = 1\nx_4 + x_5 + x_6 = 2\nEnd\n"
TWO_SIMPLEX_NON_UNIQUE = "Minimize\n1 x_1 + 1 x_2 + 1 x_3\n+2 x_4 + 1 x_5 + 1 x_6\nSubject To\nx_1 + x_2 + x_3 + x_4 = 1\nx_4 + x_5 + x_6 = 2\nEnd\n"


@pytest.mark.parametrize("precision", ["float", "double"])
def test_bdds_solution_reference_kat(precision):
    # test/test_bdd_cuda_base_sol.cpp:30-86, both blocks: bdd_cuda_base<float>, set_cost per variable, bdds_solution() as
    # two_dim_variable_array [variable][bdd] (bdd_cuda_base.cu:1204-1233; the sorted = 1 branch of bddmma_bdds_solution)
    ilp = parse_lp(TWO_SIMPLEX_DIFF_SIZE)
    s = bdd_hip_parallel_mma(to_bdd_collection(ilp), precision=precision)
    assert s.nr_variables() == 6 and s.nr_bdds() == 2
    for i, c in enumerate(ilp.objective):
        s.set_cost(c, i)
    sol = s.bdds_solution()
    assert len(sol) == 6 and [len(x) for x in sol] == [1, 1, 1, 2, 1, 1]
    # :48-56, exact values (BDD 0 has the tie x_2 / x_4 at cost 1: `cost_diff > 0 -> 0 else 1` takes the earlier variable)
    assert (sol[0][0], sol[1][0], sol[2][0], sol[3][0]) == (0, 1, 0, 0)
    assert (sol[3][1], sol[4][0], sol[5][0]) == (1, 1, 0)
    ilp = parse_lp(TWO_SIMPLEX_NON_UNIQUE)
    s = bdd_hip_parallel_mma(to_bdd_collection(ilp), precision=precision)
    for i, c in enumerate(ilp.objective):
        s.set_cost(c, i)
    sol = s.bdds_solution()
    assert [len(x) for x in sol] == [1, 1, 1, 2, 1, 1]
    assert sol[0][0] + sol[1][0] + sol[2][0] + sol[3][0] == 1      # :83-84
    assert sol[3][1] + sol[4][0] + sol[5][0] == 2


@pytest.mark.parametrize("opts", [dict(), dict(pack_width=64, waves_per_block=1), dict(wide_pack_width=64, pack_width=64)],
                         ids=["default", "narrow64", "wide_and_huge"])
def test_bdds_solution_sorted_vs_oracle(opts):
    """bdds_solution() ([variable][bdd] order) against the oracle's bdds_solution_vec (bdd_parallel_mma_base.cpp:1197-1275) on an
    instance where every variable sits in several BDDs, after a few iterations (non-trivial reparametrised costs)."""
    rng = np.random.Generator(np.random.PCG64(99))
    col = BddCollection()
    V = 60
    for _ in range(40):
        col.add_covering(np.sort(rng.choice(V, size=int(rng.integers(3, 9)), replace=False)))
    for _ in range(25):
        col.add_simplex(np.sort(rng.choice(V, size=int(rng.integers(3, 7)), replace=False)))
    for _ in range(12):
        k = int(rng.integers(8, 16))
        co = rng.integers(1, 30, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), np.sort(rng.choice(V, size=k, replace=False)))
    costs = rng.normal(0, 3, col.nr_variables())
    s = bdd_hip_parallel_mma(col, costs, precision="double", **opts)
    o = Oracle(col, costs, "double")
    for _ in range(4):
        s.iteration(); o.iteration()
    osol = o.bdds_solution_vec()
    ovar, obdd = o.layer_info()
    nb = s.get_num_bdds_per_var()
    assert nb.max() >= 3
    sol2d = s.bdds_solution()
    assert [len(x) for x in sol2d] == list(nb)
    # oracle layers regrouped as [variable][bdd ascending]
    order = np.lexsort((obdd, ovar))
    ptr = np.concatenate([[0], np.cumsum(nb)])
    for v in range(s.nr_variables()):
        np.testing.assert_array_equal(np.asarray(sol2d[v]), osol[order][ptr[v]:ptr[v + 1]], err_msg=f"variable {v}")
    # and the flat vector in internal order agrees through the layer permutation
    perm = oracle_layer_perm(s, o)
    np.testing.assert_array_equal(s.bdds_solution_vec()[perm], osol)


@pytest.mark.parametrize("P,kat", [(SHORT_CHAIN, 1.0), (LONG_CHAIN, -9.0), (GRID_3X3, -8.0), ("matching", -6.0)])
def test_200_iterations_kats_and_reparametrisation(P, kat):
    # test/test_bdd_cuda_parallel_mma.cu:197-247: 200 iteration()s, distribute_delta(), final LB,
    # primal objective vector equals the ILP objective before and after (1e-12 there, double)
    ilp = assignment_ilp(3) if P == "matching" else mrf_ilp(**P)
    col = to_bdd_collection(ilp)
    s = bdd_hip_parallel_mma(col, ilp.objective, precision="double")
    np.testing.assert_allclose(s.get_primal_objective_vector_host(), ilp.objective, atol=1e-12)
    for _ in range(200):
        s.iteration()
    s.distribute_delta()
    assert abs(s.lower_bound() - kat) < 1e-9
    np.testing.assert_allclose(s.get_primal_objective_vector_host(), ilp.objective, atol=1e-9)


# ---------------------------------------------------------------- reference traces (oracle/_ref)
@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("pack_width", [64, 128, 256])
def test_parity_protocol(name, precision, pack_width):
    """test/test_cuda_parallel_mma.cu:13-103 against traces of the reference's CPU node arithmetic."""
    col, z = load_golden(name)
    sfx = suffix(precision)
    s = bdd_hip_parallel_mma(col, precision=precision, pack_width=pack_width)
    V = s.nr_variables()
    o = Oracle(col, None, precision)
    assert V == o.nr_variables() and s.nr_bdds() == o.nr_bdds() and s.nr_layers() == o.nr_layers()
    np.testing.assert_array_equal(s.get_num_bdds_per_var(), o.nr_bdds_per_var())
    assert close(s.lower_bound(), 0.0, precision)  # before cost update
    s.update_costs([], pad_costs(z["costs"], V))
    scale = float(np.abs(z["costs"]).max())
    assert close(s.lower_bound(), float(z[f"lb_init_{sfx}"]), precision, scale)
    d = np.zeros(2 * V, s.value_type)
    t = TOL[precision]
    for it in range(10):
        s.forward_mm(0.5, d)
        np.testing.assert_allclose(d, z[f"delta_trace_{sfx}"][it, 0], atol=t["abs"] * scale, rtol=t["rel"])
        s.backward_mm(0.5, d)
        np.testing.assert_allclose(d, z[f"delta_trace_{sfx}"][it, 1], atol=t["abs"] * scale, rtol=t["rel"])
        assert close(s.lower_bound(), float(z[f"lb_trace_{sfx}"][it]), precision, scale)


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("precision", ["double", "float"])
def test_iteration_trajectory(name, precision):
    col, z = load_golden(name)
    sfx = suffix(precision)
    s = bdd_hip_parallel_mma(col, pad_costs(z["costs"], col.nr_variables()), precision=precision)
    scale = float(np.abs(z["costs"]).max())
    prev = s.lower_bound()
    for it in range(20):
        s.iteration()
        lb = s.lower_bound()
        assert close(lb, float(z[f"iter_lb_{sfx}"][it]), precision, scale)
        assert lb >= prev - TOL[precision]["abs"] * scale * 10
        prev = lb


def test_backward_mm_needs_forward_state():
    col, z = load_golden("loose_covering")
    s = bdd_hip_parallel_mma(col, z["costs"])
    d = np.zeros(2 * s.nr_variables())
    from bdd_amd.capi import BddMmaError
    with pytest.raises(BddMmaError, match="forward"):
        s.backward_mm(0.5, d)  # assert(forward_state_valid_), bdd_cuda_parallel_mma.cu:304


# ---------------------------------------------------------------- vs the oracle on seeded instances
def oracle_layer_perm(s, o):
    """internal layer order -> oracle (BDD-major) order"""
    perm = s.bdd_major_order()
    var, bdd = o.layer_info()
    np.testing.assert_array_equal(s.get_primal_variable_index()[perm], var)
    np.testing.assert_array_equal(s.get_bdd_index()[perm], bdd)
    return perm


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("pack_width,wpb,vars_per_bin", [(64, 4, 0), (128, 4, 64), (128, 1, 8192), (128, 8, 0), (128, 2, 2048)])
def test_random_cover_vs_oracle(precision, pack_width, wpb, vars_per_bin):
    # vars_per_bin 64 / 2048 / 8192: the 256- / 512- / 1024-thread exchange kernels; (128, 2, 2048) also packs into half the lanes (pack_fill)
    col, costs = random_set_cover(3000 if vars_per_bin < 2048 else 20000, 2500, 8, seed=5)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=pack_width, waves_per_block=wpb,
                             stage_cap=256 if wpb == 8 else 0, vars_per_bin=vars_per_bin, pack_fill=64 if vars_per_bin == 2048 else 0)
    o = Oracle(col, costs, precision)
    assert s.nr_packs() > 8
    assert close(s.lower_bound(), o.lower_bound(), precision, 10)
    np.testing.assert_allclose(s.lower_bound_per_bdd(), o.lower_bound_per_bdd(), rtol=TOL[precision]["rel"], atol=TOL[precision]["abs"])
    for _ in range(15):
        s.iteration(); o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 10)
    perm = oracle_layer_perm(s, o)
    # costs after 15 iterations agree layer by layer
    lo, hi, _ = s.get_solver_costs()
    olo, ohi = o.get_costs()
    np.testing.assert_allclose(lo[perm], olo, rtol=1e-4 if precision == "float" else 1e-9, atol=1e-3 if precision == "float" else 1e-9)
    np.testing.assert_allclose(hi[perm], ohi, rtol=1e-4 if precision == "float" else 1e-9, atol=1e-3 if precision == "float" else 1e-9)


def test_min_marginals_and_solution_vs_oracle():
    col, z = load_golden("knapsack_mixed")
    costs = pad_costs(z["costs"], col.nr_variables())
    s = bdd_hip_parallel_mma(col, costs, precision="double")
    o = Oracle(col, costs, "double")
    for _ in range(3):
        s.iteration(); o.iteration()
    # (no distribute_delta here: the GPU solver hands every layer back its own contribution,
    #  bdd_cuda_base.cu:1396-1436, the CPU solver spreads the averaged delta, bdd_parallel_mma_base.cpp:1046-1072;
    #  the arc costs themselves agree layer by layer)
    perm = oracle_layer_perm(s, o)
    var, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
    omm = o.min_marginals()
    np.testing.assert_allclose(mm0[perm], omm[:, 0], atol=1e-9)
    np.testing.assert_allclose(mm1[perm], omm[:, 1], atol=1e-9)
    # sorted output: ordered by (variable, bdd)
    svar, s0, s1 = s.min_marginals_cuda(get_sorted=True)
    assert np.all(np.diff(svar) >= 0)
    order = np.lexsort((s.get_bdd_index(), s.get_primal_variable_index()))
    np.testing.assert_array_equal(svar, s.get_primal_variable_index()[order])
    np.testing.assert_allclose(s0, mm0[order]); np.testing.assert_allclose(s1, mm1[order])
    # per-BDD argmin path: feasible for its BDD and attains the BDD's lower bound (test_bdd_cuda_base_sol.cpp:30-86)
    sol = s.bdds_solution_vec()
    lo, hi, _ = s.get_solver_costs()
    lbs = s.lower_bound_per_bdd()
    bdd = s.get_bdd_index(); v = s.get_primal_variable_index()
    for b in range(s.nr_bdds()):
        m = bdd == b
        x = np.zeros(s.nr_variables()); x[v[m]] = sol[m]
        assert col.evaluate(b, x)
        assert abs(np.where(sol[m] == 1, hi[m], lo[m]).sum() - lbs[b]) < 1e-9


@pytest.mark.parametrize("precision", ["double", "float"])
def test_wide_packs_vs_oracle(precision):
    rng = np.random.Generator(np.random.PCG64(21))
    col = BddCollection()
    V = 40
    for _ in range(6):
        k = int(rng.integers(16, 22))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        co = rng.integers(1, 40, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), vs)
    for _ in range(30):
        col.add_covering(np.sort(rng.choice(V, size=5, replace=False)))
    costs = rng.normal(0, 3, col.nr_variables()).round(3)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=64, wide_pack_width=512)
    o = Oracle(col, costs, precision)
    assert close(s.lower_bound(), o.lower_bound(), precision, 100)
    for _ in range(12):
        s.iteration(); o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 100)
    perm = oracle_layer_perm(s, o)
    _, mm0, mm1 = s.min_marginals_cuda(False)
    s2 = bdd_hip_parallel_mma(col, costs, precision=precision)  # fresh: compare plain min-marginals
    o2 = Oracle(col, costs, precision)
    _, a0, a1 = s2.min_marginals_cuda(False)
    om = o2.min_marginals()
    p2 = oracle_layer_perm(s2, o2)
    np.testing.assert_allclose(a0[p2], om[:, 0], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(a1[p2], om[:, 1], rtol=1e-5, atol=1e-4)
    sol = s2.bdds_solution_vec()
    for b in range(6):
        m = s2.get_bdd_index() == b
        x = np.zeros(col.nr_variables()); x[s2.get_primal_variable_index()[m]] = sol[m]
        assert col.evaluate(b, x)


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("variant", [0, 3])   # narrow + wide sweeps in one launch / as separate launches
def test_staggered_wide_packs_vs_oracle(precision, variant):
    """Wide packs that chain their BDDs (hop_root of the wide packs, k_*_wide2): a pack is one BDD wide and the next BDD starts where the
    previous one narrows.  Forced here by pack_stagger on a handful of wide BDDs; the large instances take it automatically."""
    import ctypes as C
    from bdd_amd import capi
    rng = np.random.Generator(np.random.PCG64(77))
    col = BddCollection()
    V = 60
    for _ in range(14):
        k = int(rng.integers(15, 20))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        co = rng.integers(1, 40, size=k)
        col.add_linear(co, "<=", int(co.sum() // 2), vs)
    for _ in range(40):
        col.add_covering(np.sort(rng.choice(V, size=5, replace=False)))
    costs = rng.normal(0, 3, col.nr_variables()).round(3)
    # the widest layer decides the pack width: one BDD per hop, so every further BDD of a pack has to start below the first hop
    widest = max(np.bincount(col.instr[int(col.delims[b]):int(col.delims[b + 1]) - 2, 2].astype(np.int64)).max() for b in range(14))
    wpw = int(-(-widest // 64) * 64)
    opts = dict(pack_width=64, wide_pack_width=wpw, pack_stagger=60, variant_flags=variant)
    # the layout really is staggered: fewer wide packs than wide BDDs
    h = C.c_void_p()
    o_ = capi.Options(64, wpw, 0, 0, 0, 0); o_.pack_stagger = 60
    L = capi.lib()
    capi.check(L.bddmma_layout_create(C.byref(h), np.ascontiguousarray(col.instr).ctypes.data_as(C.c_void_p),
                                      np.ascontiguousarray(col.delims).ctypes.data_as(C.c_void_p), col.nr_bdds(), C.byref(o_)), None)
    n_wide_packs = int(L.bddmma_layout_size(h, 4))
    L.bddmma_layout_destroy(h)
    n_wide_bdds = sum(1 for b in range(14) if np.bincount(col.instr[int(col.delims[b]):int(col.delims[b + 1]) - 2, 2].astype(np.int64)).max() > 64)
    assert 0 < n_wide_packs < n_wide_bdds, (n_wide_packs, n_wide_bdds)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, **opts)
    o = Oracle(col, costs, precision)
    assert close(s.lower_bound(), o.lower_bound(), precision, 100)
    np.testing.assert_allclose(s.lower_bound_per_bdd(), o.lower_bound_per_bdd(), rtol=TOL[precision]["rel"], atol=100 * TOL[precision]["abs"])
    for _ in range(12):
        s.iteration(); o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 100)
    np.testing.assert_allclose(s.lower_bound_per_bdd(), o.lower_bound_per_bdd(), rtol=TOL[precision]["rel"], atol=100 * TOL[precision]["abs"])
    perm = oracle_layer_perm(s, o)
    if precision == "double":
        lo, hi, _ = s.get_solver_costs()
        olo, ohi = o.get_costs()
        np.testing.assert_allclose(lo[perm], olo, rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(hi[perm], ohi, rtol=1e-9, atol=1e-9)
    # plain min-marginals and the argmin paths on a fresh solver (MARGINALS / SOLUTION modes of the wide kernels)
    s2 = bdd_hip_parallel_mma(col, costs, precision=precision, **opts)
    o2 = Oracle(col, costs, precision)
    _, a0, a1 = s2.min_marginals_cuda(False)
    om = o2.min_marginals()
    p2 = oracle_layer_perm(s2, o2)
    np.testing.assert_allclose(a0[p2], om[:, 0], rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(a1[p2], om[:, 1], rtol=1e-5, atol=1e-4)
    sol = s2.bdds_solution_vec()
    lo, hi, _ = s2.get_solver_costs()
    lbs = s2.lower_bound_per_bdd()
    for b in range(col.nr_bdds()):
        m = s2.get_bdd_index() == b
        x = np.zeros(col.nr_variables()); x[s2.get_primal_variable_index()[m]] = sol[m]
        assert col.evaluate(b, x)
        assert abs(np.where(sol[m] == 1, hi[m], lo[m]).sum() - lbs[b]) < (1e-9 if precision == "double" else 1e-3)


@pytest.mark.parametrize("seed,stagger,wpw", [(1, 50, 0), (2, 36, 128), (3, 0, 0), (4, 90, 256)])
def test_medium_general_rows_chained_packs_vs_oracle(seed, stagger, wpw):
    """A few thousand general linear rows of 10-20 variables with random coefficients (layers of 30-140 nodes: narrow and wide packs, both chained
    over several BDD lengths), random cover rows in between: bound, arc costs and per-BDD bounds against the CPU oracle after 8 iterations."""
    from bdd_amd import native
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    V = 6000
    rows = []
    for _ in range(2500):
        k = int(rng.integers(10, 21))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        co = rng.integers(1, 30, size=k)
        rows.append((co, vs, "<=" if rng.random() < 0.7 else ">=", int(co.sum() // 2)))
    for _ in range(4000):
        rows.append((np.ones(6, int), np.sort(rng.choice(V, size=6, replace=False)), ">=", 1))
    col = native.rows_to_bdd_collection(rows)
    costs = rng.uniform(-5, 5, col.nr_variables()).round(3)
    s = bdd_hip_parallel_mma(col, costs, precision="double", pack_stagger=stagger, wide_pack_width=wpw)
    f = bdd_hip_parallel_mma(col, costs, precision="float", pack_stagger=stagger, wide_pack_width=wpw)
    o = Oracle(col, costs, "double", threads=min(os.cpu_count() or 1, 16))
    assert abs(s.lower_bound() - o.lower_bound()) <= 1e-9 * abs(o.lower_bound())
    for _ in range(8):
        s.iteration(); f.iteration(); o.iteration()
        ref = o.lower_bound()
        assert abs(s.lower_bound() - ref) <= 1e-9 * abs(ref)
        assert abs(f.lower_bound() - ref) <= 1e-5 * abs(ref)
    perm = oracle_layer_perm(s, o)
    lo, hi, _ = s.get_solver_costs()
    olo, ohi = o.get_costs()
    np.testing.assert_allclose(lo[perm], olo, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(hi[perm], ohi, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(s.lower_bound_per_bdd(), o.lower_bound_per_bdd(), rtol=1e-9, atol=1e-9)
    sol = s.bdds_solution_vec()
    lbs = s.lower_bound_per_bdd()
    bdd = s.get_bdd_index()
    order = np.argsort(bdd, kind="stable")
    picked = np.where(sol == 1, hi, lo)[order]
    starts = np.searchsorted(bdd[order], np.arange(col.nr_bdds()))
    np.testing.assert_allclose(np.add.reduceat(picked, starts), lbs, rtol=1e-9, atol=1e-8)   # every argmin path attains its BDD's bound


def test_deterministic_mode_is_bit_reproducible_and_agrees():
    col, costs = random_set_cover(3000, 2500, 8, seed=9)
    runs = []
    for _ in range(2):
        s = bdd_hip_parallel_mma(col, costs, precision="float", deterministic=True)
        s.iterations(10)
        runs.append((s.lower_bound(), s.get_delta().copy(), s.get_solver_costs()[1].copy()))
    assert runs[0][0] == runs[1][0]
    np.testing.assert_array_equal(runs[0][1], runs[1][1])
    np.testing.assert_array_equal(runs[0][2], runs[1][2])
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    s.iterations(10)
    assert abs(s.lower_bound() - runs[0][0]) <= 1e-5 * abs(runs[0][0])
    # the default exchange keeps only the broadcast pairs; the per-variable delta is rebuilt on demand
    d = s.get_delta()
    np.testing.assert_allclose(d, runs[0][1], rtol=1e-4, atol=1e-5)
    assert np.abs(d).max() > 0
    np.testing.assert_array_equal(s.get_delta(), d)


@pytest.mark.parametrize("precision", ["float", "double"])
@pytest.mark.parametrize("opts", [dict(), dict(vars_per_bin=64), dict(vars_per_bin=1500), dict(vars_per_bin=3000, pack_width=64), dict(waves_per_block=2, pack_width=256)])
def test_deterministic_exchange_one_launch_equals_the_gathers_bit_for_bit(precision, opts):
    """`deterministic`: k_exchange_seg (one launch, fixed schedule in LDS; layout.hpp: SegExchange) and k_delta_gather + k_exchange_bcast
    (variant_flags bit 17) add the same numbers in the same (variable, bdd) order, 256 / 512 / 1024 threads per bin, with every pass and
    with the CPU oracle's order: bounds, deltas and arc costs agree bit for bit over iterations; in double the oracle's bound too."""
    from oracle.oracle import Oracle
    col, costs = random_set_cover(4000, 3500, 7, seed=11)
    one = bdd_hip_parallel_mma(col, costs, precision=precision, deterministic=True, **opts)
    two = bdd_hip_parallel_mma(col, costs, precision=precision, deterministic=True, variant_flags=0x20000, **opts)
    o = Oracle(col, costs, precision)
    for it in range(6):
        one.iteration(); two.iteration(); o.iteration()
        assert one.lower_bound() == two.lower_bound()
        np.testing.assert_array_equal(one.get_delta(), two.get_delta())
        if precision == "double":
            assert abs(one.lower_bound() - o.lower_bound()) <= 1e-11 * abs(o.lower_bound())
    for a, b in zip(one.get_solver_costs(), two.get_solver_costs()):
        np.testing.assert_array_equal(a, b)
    # explicit passes (forward_mm / backward_mm with a delta vector) and run_solver go through the same exchange
    d1, d2 = np.zeros(2 * one.nr_variables(), one.value_type), np.zeros(2 * two.nr_variables(), two.value_type)
    one.forward_mm(0.5, d1); two.forward_mm(0.5, d2)
    np.testing.assert_array_equal(d1, d2)
    from bdd_amd.solver import run_solver
    r1, r2 = run_solver(one, max_iter=25, tolerance=0.0, improvement_slope=0.0), run_solver(two, max_iter=25, tolerance=0.0, improvement_slope=0.0)
    assert r1["iterations"] == r2["iterations"] == 25 and r1["lb_final"] == r2["lb_final"]


def test_dual_ops_vs_numpy():
    col, costs = random_set_cover(500, 400, 6, seed=2)
    s = bdd_hip_parallel_mma(col, costs, precision="double")
    s.iterations(3)
    lo, hi, mm = s.get_solver_costs()
    np.testing.assert_allclose(s.net_solver_costs(), hi - lo + mm, atol=1e-12)  # bdd_cuda_parallel_mma.cu:432-446
    rng = np.random.Generator(np.random.PCG64(1))
    g = rng.normal(size=s.nr_layers())
    v = s.get_primal_variable_index()
    g2 = g.copy()
    s.make_dual_feasible(g2)
    sums = np.zeros(s.nr_variables()); np.add.at(sums, v, g)
    cnt = np.maximum(s.get_num_bdds_per_var(), 1)
    np.testing.assert_allclose(g2, g - (sums / cnt)[v], atol=1e-12)
    lb0 = s.lower_bound()
    s.gradient_step(g2, 1e-3)
    lo2, hi2, _ = s.get_solver_costs()
    np.testing.assert_allclose(hi2, hi + 1e-3 * g2, atol=1e-12)
    np.testing.assert_array_equal(lo2, lo)
    assert s.lower_bound() != lb0
    # set_solver_costs restores the state exactly
    s.set_solver_costs(lo, hi, mm)
    assert abs(s.lower_bound() - lb0) < 1e-9


def test_save_load_roundtrip():
    # test/test_bdd_cuda_base_serialization.cpp:52-76: counts and lower bound survive
    col, costs = random_set_cover(500, 400, 6, seed=3)
    s = bdd_hip_parallel_mma(col, costs, precision="double")
    s.iterations(4)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "solver.bin")
        s.save(path)
        t = bdd_hip_parallel_mma.load(path)
    assert (t.nr_variables(), t.nr_bdds(), t.nr_layers()) == (s.nr_variables(), s.nr_bdds(), s.nr_layers())
    assert t.lower_bound() == s.lower_bound()
    np.testing.assert_array_equal(t.get_delta(), s.get_delta())
    s.iterations(3); t.iterations(3)
    assert abs(t.lower_bound() - s.lower_bound()) < 1e-9


def test_checkpoint_holds_the_layout_and_rejects_corrupt_files():
    """bdd_cuda_base.cu:1486-1550 archives the layout arrays: loading does not rebuild anything, and a truncated / corrupted file is
    refused instead of being handed to the kernels (delimiters of the old format were trusted)."""
    from bdd_amd import capi
    col, costs = random_set_cover(3000, 2500, 8, seed=23)
    for kw in (dict(), dict(pack_width=64, exchange_by_variable=2), dict(resident_sweeps=1, waves_per_block=1)):
        s = bdd_hip_parallel_mma(col, costs, precision="float", **kw)
        s.iterations(6)
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "solver.bin")
            s.save(path)
            t = bdd_hip_parallel_mma.load(path)
            assert t.lower_bound() == s.lower_bound() and t.nr_packs() == s.nr_packs()
            np.testing.assert_array_equal(t.get_primal_variable_index(), s.get_primal_variable_index())
            s.iterations(5); t.iterations(5)
            assert abs(t.lower_bound() - s.lower_bound()) <= 1e-6 * abs(s.lower_bound())
            raw = open(path, "rb").read()
            bad = os.path.join(td, "bad.bin")
            for cut in (20, 200, len(raw) // 3, len(raw) - 16):                 # truncated
                open(bad, "wb").write(raw[:cut])
                with pytest.raises(capi.BddMmaError):
                    bdd_hip_parallel_mma.load(bad)
            arr = bytearray(raw)
            off = 8 + 32 + 120 + 48                                             # inside the first array records
            for pos_ in (off + 8, off + 16, len(raw) // 4, len(raw) // 2):      # corrupted: sizes / offsets no longer consistent
                b2 = bytearray(arr)
                b2[pos_:pos_ + 8] = (2**63 - 1).to_bytes(8, "little")
                open(bad, "wb").write(bytes(b2))
                with pytest.raises(capi.BddMmaError):                           # format 07: the checksum covers every byte
                    bdd_hip_parallel_mma.load(bad)


def test_checkpoint_index_arrays_are_validated_not_only_checksummed():
    """a file with a matching header must not reach the kernels (or the host-side copies) with indices, offsets or sizes
    outside their targets.  Every case below carries a CORRECT checksum, so it is the index validation that has to refuse it."""
    from bdd_amd import capi
    col, costs = random_set_cover(600, 500, 6, seed=4)
    s = bdd_hip_parallel_mma(col, costs, precision="double", waves_per_block=2)
    s.iterations(3)
    ids = CHECKPOINT_ARRAY_IDS
    with tempfile.TemporaryDirectory() as td:
        path, bad = os.path.join(td, "solver.bin"), os.path.join(td, "bad.bin")
        s.save(path)
        raw = open(path, "rb").read()
        head, sc, opts, recs, stored, tail = _parse_checkpoint(raw)
        cs = _Checksum(); cs.add(sc); cs.add(opts)
        for i, es, cnt, data in recs:
            cs.add(i.to_bytes(8, "little") + es.to_bytes(8, "little") + cnt.to_bytes(8, "little")); cs.add(data)
        assert cs.value() == stored                                     # the Python restatement of the checksum matches the library's
        _write_checkpoint(bad, head, sc, opts, recs, tail)              # ... and a rewritten, unmodified file loads
        assert bdd_hip_parallel_mma.load(bad).lower_bound() == s.lower_bound()
        # a flipped byte anywhere in the layout section is caught by the checksum
        b2 = bytearray(raw); b2[len(raw) // 3] ^= 0x40
        open(bad, "wb").write(bytes(b2))
        with pytest.raises(capi.BddMmaError, match="checksum|corrupt"):
            bdd_hip_parallel_mma.load(bad)

        def poke(name, index, value, dtype):
            out = []
            for i, es, cnt, data in recs:
                if i == ids[name]:
                    a = np.frombuffer(data, dtype=dtype).copy()
                    a[index] = value
                    data = a.tobytes()
                out.append([i, es, cnt, data])
            return out
        n_layers, n_vars = s.nr_layers(), s.nr_variables()
        cases = [
            poke("evar", 7, n_vars + 3, np.uint32), poke("bvar", 5, 60000, np.uint16), poke("lpos", 0, n_layers, np.uint32),
            poke("vpos", 3, 2**31, np.uint32), poke("var_layers", 9, n_layers + 1, np.uint32), poke("layer_var", 2, -1, np.int32),
            poke("bdd_root_slot", 1, 2**32 - 1, np.uint32), poke("cs_entry", 4, n_layers, np.uint32), poke("cs_slot", 4, 65000, np.uint16),
            poke("pack_hdr", 1, 10**6, np.uint32), poke("pack_hdr", 6, 2**31, np.uint32), poke("quad_hdr", 1, 10**6, np.uint32),
            poke("grp_hop_end", 0, 2**30, np.uint32), poke("grp_layer_off", 1, 2**30, np.uint32), poke("num_bdds_per_var", 0, -5, np.int32),
            poke("narrow_words", 0, 0x1FF | (0x1FF << 9), np.uint32),       # children 511: outside the LDS frontier of a 128-slot pack
        ]
        # per-hop statistics one entry short: bddmma_layers_per_hop copies size() elements into a caller buffer of n_hops
        short = []
        for i, es, cnt, data in recs:
            if i == ids["layers_per_hop"]:
                cnt, data = cnt - 1, data[:-8]
            short.append([i, es, cnt, data])
        cases.append(short)
        # scalars: pack_width 100, waves_per_block 3, stage_cap 0
        sc_cases = []
        for off, val in ((56, 100), (56 + 16 + 12, 3), (56 + 16 + 8, 0)):
            b = bytearray(sc); b[off:off + 4] = int(val).to_bytes(4, "little"); sc_cases.append(bytes(b))
        for k, r in enumerate(cases):
            _write_checkpoint(bad, head, sc, opts, r, tail)
            with pytest.raises(capi.BddMmaError, match="corrupt"):
                bdd_hip_parallel_mma.load(bad)
        for b in sc_cases:
            _write_checkpoint(bad, head, b, opts, recs, tail)
            with pytest.raises(capi.BddMmaError, match="corrupt"):
                bdd_hip_parallel_mma.load(bad)


def test_run_solver_and_lbfgs():
    col, costs = random_set_cover(3000, 2500, 8, seed=13)
    s = bdd_hip_parallel_mma(col, costs, precision="double")
    res = run_solver(s, max_iter=60, tolerance=1e-9, improvement_slope=0.0, time_limit=100)
    assert res["iterations"] == 60 or res["stop_reason"] == 2
    assert res["lb_final"] >= res["lb_initial"] - 1e-9
    o = Oracle(col, costs, "double")
    for _ in range(res["iterations"]):
        o.iteration()
    assert abs(res["lb_final"] - o.lower_bound()) <= 1e-9 * abs(o.lower_bound())
    # L-BFGS: lower bound never decreases (assert at lbfgs_impl.h:403) and ends at least as high as plain MMA
    s2 = bdd_hip_parallel_mma(col, costs, precision="double")
    l = bdd_hip_lbfgs(s2)
    prev = s2.lower_bound()
    for _ in range(60):
        l.iteration()
        lb = l.lower_bound()
        assert lb >= prev - 1e-6
        prev = lb
    assert prev >= res["lb_final"] - 1e-4 * abs(res["lb_final"])


# ---------------------------------------------------------------- BASELINE.json full size, size-independent properties
@pytest.mark.parametrize("tag", ["1m", "10m"])
def test_full_size_properties(tag):
    """BASELINE.json configs[1] / configs[2]: the mt19937_64(12345) instance the benchmark runs, against the lower-bound
    trajectory the reference-compiled code produced on it (tests/golden/fullsize_set_cover_mt.npz, oracle/make_golden.py
    --fullsize: reference bdd_collection + bdd_branch_instruction node arithmetic), plus size-independent properties and the
    restated oracle run here."""
    z = np.load(FULLSIZE)
    n_vars, n_rows, k, seed, iters = (int(x) for x in z[f"{tag}_params"])
    col, costs = random_set_cover_mt(n_vars, n_rows, k, seed)
    assert col.nr_bdd_nodes() == n_rows * 21
    assert abs(costs.sum() - float(z[f"{tag}_costs_sum"][0])) <= 1e-9 * costs.sum()      # same instance as the fixture's
    ref64, ref32 = z[f"{tag}_lb_f64"], z[f"{tag}_lb_f32"]
    sf = bdd_hip_parallel_mma(col, costs, precision="float")
    sd = bdd_hip_parallel_mma(col, costs, precision="double")
    assert abs(sd.lower_bound() - ref64[0]) <= 1e-9 * abs(ref64[0])
    assert abs(sf.lower_bound() - ref32[0]) <= 1e-5 * abs(ref32[0])
    lbs = []
    for it in range(iters):
        sf.iteration(); sd.iteration()
        lf, ld = sf.lower_bound(), sd.lower_bound()
        assert abs(ld - ref64[it + 1]) <= 1e-9 * abs(ref64[it + 1]), (it, ld, ref64[it + 1])   # reference node arithmetic, double
        assert abs(lf - ref32[it + 1]) <= 1e-5 * abs(ref32[it + 1]), (it, lf, ref32[it + 1])   # BASELINE.json: 1e-5 rel.
        assert abs(lf - ld) <= 1e-5 * abs(ld)            # float vs double agree (~1e-11 on CPU)
        lbs.append(ld)
    assert all(b >= a - 1e-9 * abs(a) for a, b in zip(lbs, lbs[1:]))  # MMA is monotone
    assert lbs[-1] <= costs.sum() + 1e-6                  # x = 1 is feasible for a covering problem
    # reparametrisation invariance: sum over BDDs of (hi - lo) per variable is the objective
    sd.distribute_delta()
    np.testing.assert_allclose(sd.get_primal_objective_vector_host(), costs, atol=1e-9)
    # per-BDD lower bounds add up to the lower bound (checksum of checksums)
    assert abs(sd.lower_bound_per_bdd().sum() - sd.lower_bound()) <= 1e-9 * abs(sd.lower_bound())
    # and the restated oracle itself at this size, same iteration count
    o = Oracle(col, costs, "double", threads=min(os.cpu_count() or 1, 32))
    for _ in range(iters):
        o.iteration()
    assert abs(lbs[-1] - o.lower_bound()) <= 1e-9 * abs(o.lower_bound())
    assert abs(lf - o.lower_bound()) <= 1e-5 * abs(o.lower_bound())


@pytest.mark.parametrize("seed", range(8))
def test_nontemporal_instantiations_fuzz(seed):
    """The non-temporal instantiations on random uniform-row instances (covering / simplex / at-most-one rows of 2-24 variables and a few of 70-150:
    one- and two-node layers side by side, several stage groups per pack, ragged pack ends), random packs per workgroup, bin size, BDD order,
    exchange and staging-address width — forced by variant_flags bit 20, against the oracle per iteration and in the min-marginals."""
    rng = np.random.Generator(np.random.PCG64(7000 + seed))
    V = int(rng.integers(60, 400))
    col = BddCollection()
    for _ in range(int(rng.integers(300, 1200))):
        k = int(rng.integers(2, min(V, int(rng.choice([6, 10, 24]))) + 1))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        [col.add_covering, col.add_simplex, lambda v: col.add_linear(np.ones(len(v), int), "<=", 1, v)][int(rng.integers(0, 3))](vs)
    for _ in range(int(rng.choice([0, 0, 12]))):
        (col.add_covering if rng.random() < 0.5 else col.add_simplex)(np.sort(rng.choice(V, size=int(rng.integers(70, min(V, 150))), replace=False)))
    costs = rng.normal(0, 4, col.nr_variables()).round(3)
    base = dict(pack_width=128, waves_per_block=int(rng.choice([4, 8])), resident_sweeps=1, vars_per_bin=int(rng.choice([0, 64, 256])),
                keep_bdd_order=int(rng.integers(0, 3)), deterministic=bool(rng.integers(0, 2)))
    wide = int(rng.choice([0, 0x4000]))
    ran = 0
    for precision in ("double", "float"):
        # float: the first generation (bit 12) is the one with the instantiation; double: first or second
        gen = 0x1000 if precision == "float" else int(rng.choice([0, 0x1000, 0x2000]))
        s = bdd_hip_parallel_mma(col, costs, precision=precision, variant_flags=0x40000 | 0x100000 | gen | wide, **base)
        if s.solve_sweep_kind() not in ("streaming1", "streaming2") or not s.nontemporal_loads():
            continue   # layers wider than two nodes / staggered packs in this draw: the general forms have no such instantiation
        ran += 1
        o = Oracle(col, costs, precision)
        for _ in range(int(rng.integers(3, 20))):
            s.iteration(); o.iteration()
            assert close(s.lower_bound(), o.lower_bound(), precision, 10), (base, precision, gen, wide)
        perm = oracle_layer_perm(s, o)
        _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
        omm = o.min_marginals()
        tol = dict(rtol=1e-9, atol=1e-8) if precision == "double" else dict(rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(mm0[perm], omm[:, 0], err_msg=str(base), **tol)
        np.testing.assert_allclose(mm1[perm], omm[:, 1], err_msg=str(base), **tol)
    assert ran >= 1, "no draw reached the non-temporal instantiations"


def test_twice_the_headline_size_runs_the_nontemporal_sweeps_by_rule_vs_oracle():
    """21 M nodes (V = 2 M, B = 1 M, k = 10; 1.1 / 1.6 GB of arrays): beyond 16 M slots the third generation hands over to the first / second, and
    beyond 640 MiB those run in the instantiation that loads potentials and staging tables non-temporally (SolverT::init: n12_nt) — chosen by
    rule here, not by variant_flags bit 20 as in the small bit-equality test.  Bounds against the restated oracle at this size."""
    col, costs = random_set_cover_mt(2_000_000, 1_000_000, 10, 12345)
    assert col.nr_bdd_nodes() == 21_000_000
    sf = bdd_hip_parallel_mma(col, costs, precision="float")
    sd = bdd_hip_parallel_mma(col, costs, precision="double")
    assert sf.solve_sweep_kind() == "streaming1" and sd.solve_sweep_kind() in ("streaming1", "streaming2")
    assert sf.nontemporal_loads() and sd.nontemporal_loads()
    o = Oracle(col, costs, "double", threads=min(os.cpu_count() or 1, 32))
    assert abs(sd.lower_bound() - o.lower_bound()) <= 1e-9 * abs(o.lower_bound())
    prev = sd.lower_bound()
    for _ in range(3):
        sf.iteration(); sd.iteration(); o.iteration()
        assert abs(sd.lower_bound() - o.lower_bound()) <= 1e-9 * abs(o.lower_bound())
        assert abs(sf.lower_bound() - o.lower_bound()) <= 1e-5 * abs(o.lower_bound())
        assert sd.lower_bound() >= prev - 1e-9 * abs(prev)
        prev = sd.lower_bound()
    assert abs(sd.lower_bound_per_bdd().sum() - sd.lower_bound()) <= 1e-9 * abs(sd.lower_bound())


# ---------------------------------------------------------------- general linear rows at the headline size (VERDICT r2 missing #2)
def test_full_size_knapsack_and_covering_rows_vs_oracle():
    """10 M nodes of general <= rows with non-unit coefficients (layers up to ~77 nodes) mixed with covering rows — the instance
    tools/widebench.py measures.  At this size the layout takes its automatic choices that small instances never see: staggered
    narrow packs (hop_root), the XCD-interleaved workgroup map, narrow + wide packs in one launch.  The reference's sweeps are
    width-agnostic (bdd_cuda_parallel_mma.cu:207-257), so the oracle is the same restated CPU parallel mma as everywhere else."""
    from bdd_amd import native
    rng = np.random.Generator(np.random.PCG64(1))
    n_knap, n_cover, V = 20000, 250000, 200000
    rows = []
    for _ in range(n_knap):
        vs = np.sort(rng.choice(V, size=14, replace=False))
        co = rng.integers(1, 30, size=14)
        rows.append((co, vs, "<=", int(co.sum() // 2)))
    for _ in range(n_cover):
        rows.append((np.ones(10, int), np.sort(rng.choice(V, size=10, replace=False)), ">=", 1))
    col = native.rows_to_bdd_collection(rows)
    assert col.nr_bdd_nodes() > 10_000_000
    costs = rng.uniform(-10, 10, col.nr_variables())
    sd = bdd_hip_parallel_mma(col, costs, precision="double")
    sf = bdd_hip_parallel_mma(col, costs, precision="float")
    o = Oracle(col, costs, "double", threads=min(os.cpu_count() or 1, 32))
    assert abs(sd.lower_bound() - o.lower_bound()) <= 1e-9 * abs(o.lower_bound())
    lbs = [sd.lower_bound()]
    for it in range(6):
        sd.iteration(); sf.iteration(); o.iteration()
        ld, lf, lo_ = sd.lower_bound(), sf.lower_bound(), o.lower_bound()
        assert abs(ld - lo_) <= 1e-9 * abs(lo_), (it, ld, lo_)
        assert abs(lf - lo_) <= 1e-5 * abs(lo_), (it, lf, lo_)
        lbs.append(ld)
    assert all(b >= a - 1e-9 * abs(a) for a, b in zip(lbs, lbs[1:]))
    # per-layer state, not only the bound: arc costs of every layer after the iterations against the oracle's
    perm = oracle_layer_perm(sd, o)
    lo, hi, _ = sd.get_solver_costs()
    olo, ohi = o.get_costs()
    np.testing.assert_allclose(lo[perm], olo, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(hi[perm], ohi, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(sd.lower_bound_per_bdd(), o.lower_bound_per_bdd(), rtol=1e-9, atol=1e-9)
    sd.distribute_delta()
    np.testing.assert_allclose(sd.get_primal_objective_vector_host(), costs, atol=1e-9)
    assert abs(sd.lower_bound_per_bdd().sum() - sd.lower_bound()) <= 1e-9 * abs(sd.lower_bound())


def test_full_size_wide_packs_only_vs_oracle():
    """25 000 general <= rows of 18 variables: 15.4 M nodes, layers of ~100 nodes, (almost) only wide packs — chained one BDD wide over three BDD
    lengths by the automatic rule of this size (hop_root of the wide packs).  Bound and per-BDD bounds against the CPU oracle."""
    from bdd_amd import native
    rng = np.random.Generator(np.random.PCG64(1))
    rows_n, k = 25000, 18
    V = 5 * rows_n
    rows = []
    for _ in range(rows_n):
        vs = np.sort(rng.choice(V, size=k, replace=False))
        co = rng.integers(1, 30, size=k)
        rows.append((co, vs, "<=", int(co.sum() // 2)))
    col = native.rows_to_bdd_collection(rows)
    assert col.nr_bdd_nodes() > 15_000_000
    costs = -rng.uniform(1, 10, col.nr_variables())
    sd = bdd_hip_parallel_mma(col, costs, precision="double")
    sf = bdd_hip_parallel_mma(col, costs, precision="float")
    o = Oracle(col, costs, "double", threads=min(os.cpu_count() or 1, 32))
    for it in range(4):
        sd.iteration(); sf.iteration(); o.iteration()
        ref = o.lower_bound()
        assert abs(sd.lower_bound() - ref) <= 1e-9 * abs(ref), (it, sd.lower_bound(), ref)
        assert abs(sf.lower_bound() - ref) <= 1e-5 * abs(ref), (it, sf.lower_bound(), ref)
    np.testing.assert_allclose(sd.lower_bound_per_bdd(), o.lower_bound_per_bdd(), rtol=1e-9, atol=1e-8)


# ---------------------------------------------------------------- third-generation streaming sweeps: a lane per layer
@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("wpb", [4, 8])
@pytest.mark.parametrize("shape", ["cover10", "mixed_short", "long"])
def test_lane_per_layer_sweeps_vs_oracle_and_second_generation(precision, wpb, shape):
    """k_fwd_narrow3 / k_bwd_narrow3 (kernels/narrow3.hpp) forced on small instances (streaming sweeps, records although the packs share
    none): delta per pass and bound per iteration against the oracle; arc costs, deferred differences and bound bit for bit against the
    second-generation kernels (same arithmetic in the same order); covering and simplex rows of 2-10 variables in one pack (one- and
    two-node layers side by side, ragged pack ends); packs of 70-150 hops (several stage groups per pack, staging rounds, hop-window
    refills, no resident headers)."""
    rng = np.random.Generator(np.random.PCG64({"cover10": 5, "mixed_short": 6, "long": 7}[shape]))
    col = BddCollection()
    V = 500
    if shape == "cover10":
        for _ in range(1500):
            col.add_covering(np.sort(rng.choice(V, size=10, replace=False)))
    elif shape == "mixed_short":
        for _ in range(1500):
            k = int(rng.integers(2, 11))
            (col.add_covering if rng.random() < 0.5 else col.add_simplex)(np.sort(rng.choice(V, size=k, replace=False)))
    else:
        for _ in range(300):
            k = int(rng.integers(70, 150))
            (col.add_covering if rng.random() < 0.5 else col.add_simplex)(np.sort(rng.choice(V, size=k, replace=False)))
    costs = rng.normal(0, 3, col.nr_variables()).round(3)
    # (deterministic: the exchange adds in a fixed order, so that "bit for bit" is a statement about the sweeps)
    opts = dict(precision=precision, pack_width=128, waves_per_block=wpb, resident_sweeps=1, deterministic=True)
    s3 = bdd_hip_parallel_mma(col, costs, variant_flags=0x2000, **opts)
    s2 = bdd_hip_parallel_mma(col, costs, variant_flags=0x2000 | 0x40000, **opts)
    # (eight packs per workgroup in double: the second generation's LDS exceeds 64 KiB and the first takes over — same arithmetic again)
    assert s3.solve_sweep_kind() == "streaming3" and s2.solve_sweep_kind() in ("streaming2", "streaming1")
    o = Oracle(col, costs, precision)
    assert close(s3.lower_bound(), o.lower_bound(), precision, 10)
    V = s3.nr_variables()
    d3, d2, do = (np.zeros(2 * V, s3.value_type) for _ in range(3))
    tol = dict(rtol=1e-9, atol=1e-9) if precision == "double" else dict(rtol=1e-4, atol=5e-4)
    for it in range(8):
        s3.forward_mm(0.5, d3); s2.forward_mm(0.5, d2); o.forward_mm(0.5, do)
        np.testing.assert_array_equal(d3, d2)
        np.testing.assert_allclose(d3, do, **tol)
        s3.backward_mm(0.5, d3); s2.backward_mm(0.5, d2); o.backward_mm(0.5, do)
        np.testing.assert_array_equal(d3, d2)
        np.testing.assert_allclose(d3, do, **tol)
        assert s3.lower_bound() == s2.lower_bound()
        assert close(s3.lower_bound(), o.lower_bound(), precision, 10)
    for a, b in zip(s3.get_solver_costs(), s2.get_solver_costs()):
        np.testing.assert_array_equal(a, b)
    s3.iterations(5); s2.iterations(5)
    assert s3.lower_bound() == s2.lower_bound()
    np.testing.assert_array_equal(s3.lower_bound_per_bdd(), s2.lower_bound_per_bdd())
    # ... and a fresh pair against the oracle's own iteration()
    s = bdd_hip_parallel_mma(col, costs, variant_flags=0x2000, **opts)
    o = Oracle(col, costs, precision)
    for _ in range(6):
        s.iteration(); o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 10)


@pytest.mark.parametrize("seed", range(16))
def test_lane_per_layer_sweeps_fuzz(seed):
    """Differential fuzz aimed at the third-generation sweeps (the general fuzz below rarely meets their conditions): covering / simplex /
    at-most-one rows of 2-24 variables (packs with one and with several stage groups), random options that keep them selected — packs per
    workgroup, bin size, BDD order, deterministic exchange, 64-bit staging addresses — both precisions against the oracle: bound per
    iteration, min-marginals, the L-BFGS view of the backward sweep (net_solver_costs) and a primal from the argmin paths."""
    rng = np.random.Generator(np.random.PCG64(5000 + seed))
    V = int(rng.integers(40, 400))
    kmax = int(rng.choice([6, 10, 24]))
    col = BddCollection()
    for _ in range(int(rng.integers(300, 1500))):
        k = int(rng.integers(2, min(V, kmax) + 1))
        vs = np.sort(rng.choice(V, size=k, replace=False))
        kind = int(rng.integers(0, 3))
        if kind == 0:
            col.add_covering(vs)
        elif kind == 1:
            col.add_simplex(vs)
        else:
            col.add_linear(np.ones(k, int), "<=", 1, vs)
    costs = rng.normal(0, 4, col.nr_variables()).round(3)
    opts = dict(pack_width=128, waves_per_block=int(rng.choice([4, 8])), resident_sweeps=1, vars_per_bin=int(rng.choice([0, 64, 256])),
                keep_bdd_order=int(rng.integers(0, 3)), deterministic=bool(rng.integers(0, 2)),
                variant_flags=0x2000 | int(rng.choice([0, 0x4000])))
    for precision in ("double", "float"):
        s = bdd_hip_parallel_mma(col, costs, precision=precision, **opts)
        assert s.solve_sweep_kind() == "streaming3", (opts, s.solve_sweep_kind())
        o = Oracle(col, costs, precision)
        n_it = int(rng.integers(3, 25))
        for _ in range(n_it):
            s.iteration(); o.iteration()
            assert close(s.lower_bound(), o.lower_bound(), precision, 10), (opts, precision)
        perm = oracle_layer_perm(s, o)
        _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
        omm = o.min_marginals()
        tol = dict(rtol=1e-9, atol=1e-8) if precision == "double" else dict(rtol=1e-4, atol=2e-3)
        np.testing.assert_allclose(mm0[perm], omm[:, 0], err_msg=str(opts), **tol)
        np.testing.assert_allclose(mm1[perm], omm[:, 1], err_msg=str(opts), **tol)
        lo, hi, mm = s.get_solver_costs()
        np.testing.assert_allclose(s.net_solver_costs(), (hi - lo) + mm, rtol=0, atol=1e-12 if precision == "double" else 1e-5)
        sol = s.bdds_solution_vec()
        bdd, v = s.get_bdd_index(), s.get_primal_variable_index()
        for b in range(min(5, col.nr_bdds())):
            m = bdd == b
            x = np.zeros(col.nr_variables()); x[v[m]] = sol[m]
            assert col.evaluate(b, x)


def test_lane_per_layer_sweeps_are_the_rule_where_they_apply():
    """The automatic choice: a uniform family at a streaming size takes the third generation in both precisions; rows with layers wider than
    two nodes, 64-slot packs and variant_flags bit 18 do not."""
    col, costs = random_set_cover(200_000, 140_000, 8, seed=3)   # 2 188 packs of 128 slots (fewer than 2 048 packs: 64-slot packs)
    for precision in ("float", "double"):
        s = bdd_hip_parallel_mma(col, costs, precision=precision, resident_sweeps=1)
        assert s.solve_sweep_kind() == "streaming3", s.solve_sweep_kind()
        o = Oracle(col, costs, precision)
        s.iterations(3)
        for _ in range(3):
            o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 10)
    assert bdd_hip_parallel_mma(col, costs, resident_sweeps=1, variant_flags=0x40000).solve_sweep_kind() == "streaming2"
    assert bdd_hip_parallel_mma(col, costs, resident_sweeps=1, pack_width=64).solve_sweep_kind() in ("streaming2", "streaming1")


@pytest.mark.parametrize("precision,flags,wpb,kind", [("float", 0x41000, 4, "streaming1"), ("float", 0x41000, 8, "streaming1"), ("double", 0x41000, 4, "streaming1"),
                                                      ("double", 0x40000, 8, "streaming1"), ("double", 0x40000, 4, "streaming2"),
                                                      ("float", 0x41000 | 0x4000, 8, "streaming1"), ("double", 0x40000 | 0x4000, 4, "streaming2")])   # bit 14: 64-bit staging addresses
def test_nontemporal_instantiations_of_the_first_and_second_generation(precision, flags, wpb, kind):
    """Beyond 640 MiB of arrays the first- and (double) second-generation streaming sweeps of 128-slot packs run in the instantiation that loads
    potentials and staging tables non-temporally (variant_flags bit 20 selects it on a small instance): a cache policy, so every result is
    bit-equal to the default instantiation's; a fresh solver agrees with the oracle.  (Double with eight packs per workgroup: the second
    generation's LDS exceeds 64 KiB and the first takes over, as at 21 M nodes and beyond.)  64-slot packs and the float second generation
    have no such instantiation and report so."""
    col, costs = random_set_cover(3000, 2400, 9, seed=17)
    opts = dict(precision=precision, pack_width=128, waves_per_block=wpb, resident_sweeps=1, deterministic=True)
    n = bdd_hip_parallel_mma(col, costs, variant_flags=flags | 0x100000, **opts)
    c = bdd_hip_parallel_mma(col, costs, variant_flags=flags, **opts)
    assert n.solve_sweep_kind() == c.solve_sweep_kind() == kind
    assert n.nontemporal_loads() and not c.nontemporal_loads()
    V = n.nr_variables()
    dn, dc = (np.zeros(2 * V, n.value_type) for _ in range(2))
    for _ in range(4):
        n.forward_mm(0.5, dn); c.forward_mm(0.5, dc)
        np.testing.assert_array_equal(dn, dc)
        n.backward_mm(0.5, dn); c.backward_mm(0.5, dc)
        np.testing.assert_array_equal(dn, dc)
        assert n.lower_bound() == c.lower_bound()
    n.iterations(6); c.iterations(6)
    for a, b in zip(n.get_solver_costs(), c.get_solver_costs()):
        np.testing.assert_array_equal(a, b)
    assert n.lower_bound() == c.lower_bound()
    s = bdd_hip_parallel_mma(col, costs, variant_flags=flags | 0x100000, **opts)
    o = Oracle(col, costs, precision)
    for _ in range(6):
        s.iteration(); o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 10)
    assert not bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=64, resident_sweeps=1, variant_flags=flags | 0x100000).nontemporal_loads()
    assert not bdd_hip_parallel_mma(col, costs, precision="float", pack_width=128, waves_per_block=4, resident_sweeps=1, variant_flags=0x40000 | 0x100000).nontemporal_loads()


# ---------------------------------------------------------------- long BDDs: hop-window refills and several stage groups per pack
@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("variant", [0, 0x2000])   # 0x2000: the second-generation streaming sweeps (per-lane records) although these packs share none
@pytest.mark.parametrize("pack_width,stage_cap,wpb", [(64, 64, 4), (128, 640, 1), (256, 256, 2), (64, 128, 8),
                                                      (256, 640, 4)])   # last: > 64 KiB of LDS per workgroup in double
def test_long_bdds_vs_oracle(precision, pack_width, stage_cap, wpb, variant):
    rng = np.random.Generator(np.random.PCG64(33))
    V = 700
    col = BddCollection()
    col.add_simplex(np.sort(rng.choice(V, size=300, replace=False)))        # 300 hops
    col.add_covering(np.sort(rng.choice(V, size=200, replace=False)))       # 200 hops
    col.add_cardinality(np.sort(rng.choice(V, size=90, replace=False)).tolist(), 3)
    for _ in range(150):                                                    # plus many short rows in the same packs
        k = int(rng.integers(2, 12))
        col.add_covering(np.sort(rng.choice(V, size=k, replace=False)))
    for _ in range(40):
        k = int(rng.integers(65, 140))                                      # longer than one 64-entry hop window
        col.add_simplex(np.sort(rng.choice(V, size=k, replace=False)))
    costs = rng.normal(0, 2, col.nr_variables()).round(3)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=pack_width, stage_cap=stage_cap, waves_per_block=wpb, variant_flags=variant)
    o = Oracle(col, costs, precision)
    assert s.nr_hops() == 300
    assert close(s.lower_bound(), o.lower_bound(), precision, 10)
    for _ in range(12):
        s.iteration(); o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 10)
    perm = oracle_layer_perm(s, o)
    _, a0, a1 = s.min_marginals_cuda(False)
    om = o.min_marginals()
    tol = dict(rtol=1e-9, atol=1e-9) if precision == "double" else dict(rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(a0[perm], om[:, 0], **tol)
    np.testing.assert_allclose(a1[perm], om[:, 1], **tol)
    sol = s.bdds_solution_vec()
    bdd, v = s.get_bdd_index(), s.get_primal_variable_index()
    for b in (0, 1, 2):
        m = bdd == b
        x = np.zeros(col.nr_variables()); x[v[m]] = sol[m]
        assert col.evaluate(b, x)


@pytest.mark.parametrize("precision", ["double", "float"])
@pytest.mark.parametrize("wpb", [0, 2, 4])
def test_resident_sweeps_with_up_to_sixty_hops(precision, wpb):
    """Rows of 17-60 variables on a small instance: resident packs of more than 16 hops (k_fwd_res2 / k_bwd_res2 walk them in blocks of 16
    with a ring of eight prefetched records), one / two / four packs per workgroup, against the first generation and the oracle."""
    rng = np.random.Generator(np.random.PCG64(61))
    V = 900
    col = BddCollection()
    for k in (17, 24, 31, 33, 47, 60):
        for _ in range(40):
            vs = np.sort(rng.choice(V, size=k, replace=False))
            (col.add_covering if rng.random() < 0.5 else col.add_simplex)(vs)
    costs = rng.normal(0, 2, col.nr_variables()).round(3)
    s = bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=64, resident_sweeps=2, waves_per_block=wpb)
    g1 = bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=64, resident_sweeps=2, waves_per_block=wpb, variant_flags=0x800)
    o = Oracle(col, costs, precision)
    assert s.nr_hops() == 60
    for it in range(15):
        s.iteration(); g1.iteration(); o.iteration()
        assert close(s.lower_bound(), o.lower_bound(), precision, 10), it
        assert close(g1.lower_bound(), o.lower_bound(), precision, 10), it
    lo, hi, mm = s.get_solver_costs()
    lo1, hi1, mm1 = g1.get_solver_costs()
    tol = dict(rtol=1e-9, atol=1e-9) if precision == "double" else dict(rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(lo, lo1, **tol)
    np.testing.assert_allclose(hi, hi1, **tol)
    np.testing.assert_allclose(mm, mm1, **tol)
    np.testing.assert_allclose(s.lower_bound_per_bdd(), o.lower_bound_per_bdd(), rtol=TOL[precision]["rel"], atol=10 * TOL[precision]["abs"])


@pytest.mark.parametrize("precision", ["double", "float"])
def test_huge_layers_use_the_global_memory_frontier(precision):
    """BDD layers wider than the LDS frontier of a workgroup (wide_pack_width, default 2048): the 'huge' pack path.
    Same protocol as the other oracle comparisons: lower bound, iterations, min-marginals, per-BDD argmin."""
    from bdd_amd import native
    rng = np.random.Generator(np.random.PCG64(1))
    n = 28
    co = rng.integers(1, 5000, size=n)
    rows = [(co, np.arange(n), "<=", int(co.sum() // 2))]
    for _ in range(6):                                   # narrow and wide neighbours sharing the variables
        k = int(rng.integers(3, 9))
        rows.append((np.ones(k, int), np.sort(rng.choice(n, size=k, replace=False)), ">=", 1))
    c2 = rng.integers(1, 40, size=n)
    rows.append((c2, np.arange(n), ">=", int(c2.sum() // 3)))
    col = native.rows_to_bdd_collection(rows)
    assert max(col.layer_widths(0)) > 2048
    costs = rng.normal(0, 5, n).round(3)
    s = bdd_hip_parallel_mma(col, costs, precision=precision)
    o = Oracle(col, costs, precision)
    rel = 1e-9 if precision == "double" else 2e-5
    assert abs(s.lower_bound() - o.lower_bound()) <= rel * max(1.0, abs(o.lower_bound()))
    for _ in range(25):
        s.iteration()
        o.iteration()
    lb, ref = s.lower_bound(), o.lower_bound()
    assert abs(lb - ref) <= rel * max(1.0, abs(ref)), (lb, ref)
    perm = oracle_layer_perm(s, o)
    _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
    omm = o.min_marginals()
    tol = dict(rtol=1e-4, atol=2e-2) if precision == "float" else dict(rtol=1e-9, atol=1e-8)
    np.testing.assert_allclose(mm0[perm], omm[:, 0], **tol)
    np.testing.assert_allclose(mm1[perm], omm[:, 1], **tol)
    sol = s.bdds_solution_vec()
    bdd, var = s.get_bdd_index(), s.get_primal_variable_index()
    for b in range(s.nr_bdds()):
        x = np.zeros(s.nr_variables()); x[var[bdd == b]] = sol[bdd == b]
        assert col.evaluate(b, x)
    # forcing the same instance through small LDS frontiers (more huge packs) gives the same numbers
    s2 = bdd_hip_parallel_mma(col, costs, precision=precision, pack_width=64, wide_pack_width=64)
    for _ in range(25):
        s2.iteration()
    assert abs(s2.lower_bound() - ref) <= rel * max(1.0, abs(ref))


@pytest.mark.parametrize("precision", ["double", "float"])
def test_edge_cases_vs_oracle(precision):
    """Inputs at the edges of the format: duplicated rows, variables no BDD contains, zero costs, unsorted variables
    with negative coefficients, a cost vector shorter than the number of variables (bdd_cuda_base.cu:465-469)."""
    col = BddCollection()
    col.add_simplex([0, 1, 2, 3, 4, 5])
    col.add_covering([2, 3, 9])
    col.add_covering([2, 3, 9])                            # duplicate row
    col.add_linear([2, -1, 3, -2], ">=", 1, [9, 4, 11, 0])  # unsorted variables, negative coefficients; variables 6, 7, 8, 10 are in no BDD
    col.add_covering([1, 5])
    V = col.nr_variables()
    assert V == 12
    rng = np.random.Generator(np.random.PCG64(5))
    costs = rng.normal(0, 3, V).round(2)
    costs[[1, 9]] = 0.0
    rel = 1e-9 if precision == "double" else 2e-5
    for c in (costs, costs[:8]):                           # the short vector leaves the tail variables at cost 0
        s = bdd_hip_parallel_mma(col, c, precision=precision)
        o = Oracle(col, pad_costs(c, V), precision)
        assert s.nr_bdds(6) == 0 and s.nr_bdds(2) == 3
        assert abs(s.lower_bound() - o.lower_bound()) <= rel * max(1.0, abs(o.lower_bound()))
        for it in range(30):
            s.iteration(); o.iteration()
            lb, ref = s.lower_bound(), o.lower_bound()
            assert np.isfinite(ref) and abs(lb - ref) <= rel * max(1.0, abs(ref)), (it, lb, ref)
        perm = oracle_layer_perm(s, o)
        _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
        omm = o.min_marginals()
        tol = dict(rtol=1e-4, atol=1e-4) if precision == "float" else dict(rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(mm0[perm], omm[:, 0], **tol)
        np.testing.assert_allclose(mm1[perm], omm[:, 1], **tol)


@pytest.mark.parametrize("precision", ["double", "float"])
def test_forced_variables_keep_the_bound_finite_and_valid(precision):
    """Single-variable BDDs and rows that fix variables: one arc of every node of such a layer leads to the bot sink, the
    min-marginal of that side is +inf and the GPU rule applies no update (`mm = 0` unless both are finite,
    bdd_cuda_parallel_mma.cu:83-84).  The CPU solver instead propagates the fixation as infinite costs
    (bdd_parallel_mma_base.cpp:844-863), so this is checked by properties, not against the oracle: the bound stays
    finite, never decreases and never exceeds the optimum."""
    from bdd_amd.ilp import ILP
    ilp = ILP()
    names = [f"x{i}" for i in range(10)]
    for n in names:
        ilp.var(n)
    ilp.objective = [3.0, -1.5, 2.0, 0.5, -2.5, 1.0, -0.5, 4.0, -3.0, 0.25]
    ilp.add_constraint([(1, "x3")], "=", 1)                                # single-variable simplex: x3 = 1
    ilp.add_constraint([(1, "x7")], ">=", 1)                               # single-variable covering
    ilp.add_constraint([(1, "x0"), (1, "x1")], "=", 2)                     # both forced to 1
    ilp.add_constraint([(1, "x4"), (1, "x5")], "<=", 0)                    # both forced to 0
    ilp.add_constraint([(1, n) for n in ("x0", "x2", "x4", "x6", "x8")], "<=", 3)
    ilp.add_constraint([(1, n) for n in ("x1", "x3", "x5", "x7", "x9")], ">=", 3)
    ilp.add_constraint([(2, "x2"), (-1, "x6"), (1, "x8"), (1, "x9")], ">=", 1)
    opt = brute_force_optimum(ilp)
    col = to_bdd_collection(ilp)
    s = bdd_hip_parallel_mma(col, ilp.objective, precision=precision)
    prev = s.lower_bound()
    assert np.isfinite(prev) and prev <= opt + 1e-6
    for _ in range(200):
        s.iteration()
        lb = s.lower_bound()
        assert np.isfinite(lb) and lb >= prev - (1e-9 if precision == "double" else 1e-4) and lb <= opt + 1e-4
        prev = lb
    _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
    var = s.get_primal_variable_index()
    assert np.isinf(mm0[var == 3]).sum() == 1 and np.all(np.isfinite(mm1[var == 3]))   # x3 = 0 is impossible in its own single-node BDD only


def test_independent_handles_interleaved_and_bad_device():
    """Handles own their stream and buffers (bdd_mma.h: one per problem, one per GPU host thread): two solvers that are
    stepped alternately, and from two host threads, give exactly the results of running each alone."""
    import threading
    col_a, costs_a = random_set_cover(3000, 2000, 6, seed=3)
    col_b, costs_b = random_set_cover(2500, 1800, 9, seed=4)

    def alone(col, costs, n):
        s = bdd_hip_parallel_mma(col, costs, precision="double", deterministic=True)
        s.iterations(n)
        return s.lower_bound()
    want_a, want_b = alone(col_a, costs_a, 40), alone(col_b, costs_b, 25)
    a = bdd_hip_parallel_mma(col_a, costs_a, precision="double", deterministic=True)
    b = bdd_hip_parallel_mma(col_b, costs_b, precision="double", deterministic=True)
    for i in range(40):
        a.iteration()
        if i < 25:
            b.iteration()
    assert a.lower_bound() == want_a and b.lower_bound() == want_b
    out = {}

    def work(key, col, costs, n):
        out[key] = alone(col, costs, n)
    ts = [threading.Thread(target=work, args=("a", col_a, costs_a, 40)), threading.Thread(target=work, args=("b", col_b, costs_b, 25))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out == {"a": want_a, "b": want_b}
    with pytest.raises(Exception, match="device"):
        bdd_hip_parallel_mma(col_a, costs_a, device=63)


@pytest.mark.parametrize("seed", range(int(os.environ.get("BDDMMA_FUZZ_SEEDS", "16"))))
def test_randomised_instances_and_layout_options_vs_oracle(seed):
    """Differential test: random mixtures of covering / simplex / cardinality / knapsack rows (no row forces a variable, so
    all min-marginals stay finite and the CPU and GPU update rules coincide) under random layout options — pack widths,
    packs per workgroup, stage groups, bin sizes, LDS-or-global frontier, BDD order kept or grouped by shape, streaming or resident
    sweeps, binned or (variable, bdd) entry order (BDDMMA_FUZZ_SEEDS widens the seed range; 400 seeds ran green before commit)."""
    from bdd_amd import native
    rng = np.random.Generator(np.random.PCG64(1000 + seed))
    V = int(rng.integers(30, 400))
    rows = []
    for _ in range(int(rng.integers(20, 300))):
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
            rows.append((co, vs, "<=", int(rng.integers(co.max(), co.sum()))))          # every variable may still be 1
        else:
            co = rng.integers(1, 9, size=k)
            rows.append((co, vs, ">=", int(rng.integers(1, co.sum() - co.max() + 1))))  # every variable may still be 0
    if seed % 2 == 1:      # a few rows with layers wider than a wavefront: workgroup-per-pack (and, with wide_pack_width 64, huge) packs;
        # up to eight of them, so that pack_stagger also chains wide BDDs (hop_root of the wide packs)
        for _ in range(int(rng.integers(1, 9))):
            k = int(rng.integers(12, min(V, 20) + 1))
            co = rng.integers(1, 60, size=k)
            rows.append((co, np.sort(rng.choice(V, size=k, replace=False)), "<=", int(rng.integers(co.max(), co.sum()))))
    col = native.rows_to_bdd_collection(rows)
    if col.nr_bdds() == 0:
        pytest.skip("all rows trivial")
    Vc = col.nr_variables()
    costs = rng.normal(0, 4, Vc).round(3)
    pw = int(rng.choice([64, 128, 256]))
    wpb = int(rng.choice([1, 2, 4, 8]))
    cap = int(rng.choice([pw, 256, 640])) if wpb < 8 else 256
    opts = dict(pack_width=pw, waves_per_block=wpb, stage_cap=max(cap, pw), vars_per_bin=int(rng.choice([0, 64, 256])),
                wide_pack_width=int(rng.choice([0, 64, 128, 256])), keep_bdd_order=int(rng.integers(0, 3)),
                resident_sweeps=int(rng.choice([0, 1, 2])), exchange_by_variable=int(rng.choice([0, 0, 2])),
                variant_flags=int(rng.choice([0, 0, 1, 2, 3])) | int(rng.choice([0, 0x800, 0x1000, 0x2000, 0x2000])) | int(rng.choice([0, 0, 0x4000])) | int(rng.choice([0, 0, 0x40000]))   # 0x800 / 0x1000: first-generation resident / streaming sweeps, 0x2000: per-lane records also for unshared packs (general form), 0x4000: 64-bit staging addresses (arrays >= 4 GiB)
                | int(os.environ.get("BDDMMA_FUZZ_VARIANT_OR", "0")),  # the env: tools/soak.sh bisections
                pack_fill=int(rng.choice([0, 0, pw // 2, 16])), pack_stagger=int(rng.choice([0, 1, 24, 60, 200])))
    s = bdd_hip_parallel_mma(col, costs, precision="double", **opts)
    o = Oracle(col, costs, "double")
    assert abs(s.lower_bound() - o.lower_bound()) <= 1e-9 * max(1.0, abs(o.lower_bound())), opts
    n_it = int(rng.integers(3, 30))
    for _ in range(n_it):
        s.iteration(); o.iteration()
    lb, ref = s.lower_bound(), o.lower_bound()
    assert np.isfinite(ref) and abs(lb - ref) <= 1e-9 * max(1.0, abs(ref)), (opts, lb, ref)
    perm = oracle_layer_perm(s, o)
    _, mm0, mm1 = s.min_marginals_cuda(get_sorted=False)
    omm = o.min_marginals()
    np.testing.assert_allclose(mm0[perm], omm[:, 0], rtol=1e-9, atol=1e-8, err_msg=str(opts))
    np.testing.assert_allclose(mm1[perm], omm[:, 1], rtol=1e-9, atol=1e-8, err_msg=str(opts))
    f = bdd_hip_parallel_mma(col, costs, precision="float", **opts)
    f.iterations(n_it)
    assert abs(f.lower_bound() - ref) <= 2e-5 * max(1.0, abs(ref)), opts


def test_handles_do_not_leak_device_memory_and_long_runs_stay_monotone():
    import torch
    col, costs = random_set_cover(20_000, 12_000, 7, seed=21)
    s = bdd_hip_parallel_mma(col, costs, precision="float")   # warm-up: code objects, allocator pools
    s.iterations(5); s.lower_bound(); s.close()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for i in range(60):
        s = bdd_hip_parallel_mma(col, costs, precision="float" if i % 2 else "double")
        if i % 3 == 0:
            l = bdd_hip_lbfgs(s)
            for _ in range(7):
                l.iteration()
            l.close()
        else:
            s.iterations(3)
        s.lower_bound()
        s.close()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)            # nothing but allocator slack is kept
    # a long run: the bound never decreases and stays finite (float accumulates 20 000 cost updates per layer)
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    prev = s.lower_bound()
    for _ in range(20):
        s.iterations(1000)
        lb = s.lower_bound()
        assert np.isfinite(lb) and lb >= prev - 1e-4 * abs(prev)
        prev = lb
    assert prev <= costs.sum()
