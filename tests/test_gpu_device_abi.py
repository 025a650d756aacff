"""The device-pointer branch of the C-ABI (`on_device = 1`) — needs an MI355X.

These are the thrust::device_vector overloads the reference's own callers use (test/test_cuda_parallel_mma.cu:71-99,
lbfgs.h:65-67, incremental_mm_agreement_rounding_cuda.cu:262-362).  Buffers are torch CUDA tensors; only their raw
data_ptr() crosses the ABI.  Every call is compared with the same call on host buffers (`on_device = 0`) and, for the
forward_mm / backward_mm protocol, with the golden traces of the reference's node arithmetic.
"""
import numpy as np
import pytest

from bdd_amd.instances import random_set_cover
from bdd_amd.solver import bdd_hip_parallel_mma
from util import GOLDEN, load_golden, pad_costs, suffix

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def tdtype(s):
    return torch.float64 if s.value_type == np.float64 else torch.float32


def dev(a):
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    torch.cuda.synchronize()
    return t


def host(s, t):
    s.synchronize()          # device-buffer calls are asynchronous on the handle's stream
    torch.cuda.synchronize()
    return t.cpu().numpy()


@pytest.mark.parametrize("name", GOLDEN)
@pytest.mark.parametrize("precision", ["double", "float"])
def test_parity_protocol_with_device_delta(name, precision):
    """test/test_cuda_parallel_mma.cu:13-103 with delta_lo_hi as a device vector (as the reference's test passes it),
    against the golden traces of the reference's CPU node arithmetic (same protocol as test_gpu_parity.test_parity_protocol)."""
    col, z = load_golden(name)
    sfx = suffix(precision)
    s = bdd_hip_parallel_mma(col, precision=precision)
    V = s.nr_variables()
    c = pad_costs(z["costs"], V).astype(s.value_type)
    s.update_costs(None, dev(c))
    scale = float(np.abs(z["costs"]).max())
    tol = dict(atol=1e-9 * scale, rtol=1e-9) if precision == "double" else dict(atol=2e-4 * scale, rtol=1e-5)
    assert abs(s.lower_bound() - float(z[f"lb_init_{sfx}"])) <= tol["atol"] + tol["rtol"] * abs(float(z[f"lb_init_{sfx}"]))
    d = torch.zeros(2 * V, dtype=tdtype(s), device="cuda")
    torch.cuda.synchronize()
    for it in range(10):
        s.forward_mm(0.5, d)
        np.testing.assert_allclose(host(s, d), z[f"delta_trace_{sfx}"][it, 0], **tol)
        s.backward_mm(0.5, d)
        np.testing.assert_allclose(host(s, d), z[f"delta_trace_{sfx}"][it, 1], **tol)
        ref = float(z[f"lb_trace_{sfx}"][it])
        assert abs(s.lower_bound() - ref) <= tol["atol"] + tol["rtol"] * abs(ref)


@pytest.mark.parametrize("precision", ["double", "float"])
def test_every_device_overload_matches_its_host_twin(precision):
    col, costs = random_set_cover(2000, 1700, 7, seed=3)
    a = bdd_hip_parallel_mma(col, precision=precision)       # driven through host buffers
    b = bdd_hip_parallel_mma(col, precision=precision)       # driven through device buffers
    T = tdtype(a)
    nv, nl, nb = a.nr_variables(), a.nr_layers(), a.nr_bdds()
    rng = np.random.Generator(np.random.PCG64(8))

    # update_costs(device_vector, device_vector): REAL elements of the handle's precision (bdd_cuda_base.cu:476-500)
    c_hi = costs.astype(a.value_type)
    c_lo = rng.uniform(0, 0.1, nv).astype(a.value_type)
    a._ck(a._L.bddmma_update_costs(a._h, c_lo.ctypes.data, nv, c_hi.ctypes.data, nv, a._prec, 0))
    b.update_costs(dev(c_lo), dev(c_hi))
    for x, y in zip(a.get_solver_costs(), b.get_solver_costs()):
        np.testing.assert_array_equal(x, y)
    # shorter device vector: tail layers are SET to 0 (bdd_cuda_base.cu:465-469), one side only
    half = nv // 2
    a._ck(a._L.bddmma_update_costs(a._h, None, 0, c_hi[:half].ctypes.data, half, a._prec, 0))
    b.update_costs(None, dev(c_hi[:half]))
    for x, y in zip(a.get_solver_costs(), b.get_solver_costs()):
        np.testing.assert_array_equal(x, y)

    # forward_mm / backward_mm on a device delta, interleaved with iteration()s
    da = np.zeros(2 * nv, a.value_type)
    db = torch.zeros(2 * nv, dtype=T, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        a.forward_mm(0.5, da); b.forward_mm(0.5, db)
        np.testing.assert_allclose(da, host(b, db), rtol=1e-6 if precision == "float" else 1e-12, atol=1e-6 if precision == "float" else 1e-12)
        a.normalize_delta(da); b.normalize_delta(db)
        a.backward_mm(0.5, da); b.backward_mm(0.5, db)
        np.testing.assert_allclose(da, host(b, db), rtol=1e-6 if precision == "float" else 1e-12, atol=1e-6 if precision == "float" else 1e-12)
        a.normalize_delta(da); b.normalize_delta(db)
        db.copy_(torch.from_numpy(da).cuda())   # keep the two runs on the same inputs (the LDS accumulation order is free)
        torch.cuda.synchronize()
    assert abs(a.lower_bound() - b.lower_bound()) <= 1e-6 * abs(a.lower_bound())
    b.set_solver_costs(*a.get_solver_costs())
    a.iterations(3); b.iterations(3)

    # get / set_solver_costs, get / set_delta
    lo, hi, mm = a.get_solver_costs()
    out = tuple(torch.empty(nl, dtype=T, device="cuda") for _ in range(3))
    a.get_solver_costs(out=out)
    for x, y in zip((lo, hi, mm), out):
        np.testing.assert_array_equal(x, host(a, y))
    b.set_solver_costs(*out)
    for x, y in zip((lo, hi, mm), b.get_solver_costs()):
        np.testing.assert_array_equal(x, y)
    d = a.get_delta()
    dt = torch.empty(2 * nv, dtype=T, device="cuda")
    np.testing.assert_array_equal(host(a, a.get_delta(out=dt)), d)
    b.set_delta(dt)
    np.testing.assert_array_equal(b.get_delta(), d)
    assert a.lower_bound() == b.lower_bound()

    # lower_bound_per_bdd, primal objective vector
    np.testing.assert_array_equal(host(a, a.lower_bound_per_bdd(out=torch.empty(nb, dtype=T, device="cuda"))), a.lower_bound_per_bdd())
    np.testing.assert_array_equal(host(a, a.get_primal_objective_vector(torch.empty(nv, dtype=T, device="cuda"))),
                                  a.get_primal_objective_vector_host())

    # min_marginals_cuda (sorted and unsorted), bdds_solution_vec
    for srt in (True, False):
        v, m0, m1 = a.min_marginals_cuda(get_sorted=srt)
        o = (torch.empty(nl, dtype=torch.int32, device="cuda"), torch.empty(nl, dtype=T, device="cuda"), torch.empty(nl, dtype=T, device="cuda"))
        a.min_marginals_cuda(get_sorted=srt, out=o)
        np.testing.assert_array_equal(host(a, o[0]), v)
        np.testing.assert_array_equal(host(a, o[1]), m0)
        np.testing.assert_array_equal(host(a, o[2]), m1)
    sol = a.bdds_solution_vec()
    np.testing.assert_array_equal(host(a, a.bdds_solution_vec(out=torch.empty(nl, dtype=torch.int8, device="cuda"))), sol)

    # L-BFGS support ops: net_solver_costs, make_dual_feasible, gradient_step
    x = a.net_solver_costs()
    np.testing.assert_array_equal(host(a, a.net_solver_costs(out=torch.empty(nl, dtype=T, device="cuda"))), x)
    g = rng.normal(size=nl).astype(a.value_type)
    gh = g.copy(); a.make_dual_feasible(gh)
    gd = dev(g); a.make_dual_feasible(gd)
    np.testing.assert_array_equal(host(a, gd), gh)
    a.gradient_step(gh, 1e-3); b.gradient_step(gd, 1e-3)
    for x, y in zip(a.get_solver_costs(), b.get_solver_costs()):
        np.testing.assert_array_equal(x, y)
    assert a.lower_bound() == b.lower_bound()


def test_device_chip_query_and_allocated_bytes():
    """bddmma_device_chip = the hipDeviceProp figures the layout rules and the input stage's split-length rule read (the reference:
    cudaGetDeviceProperties, bdd_preprocessor.cpp:21-30); bddmma_device_allocated_bytes >= bddmma_device_bytes, and a tiny solver no
    longer holds a 32 MiB arena."""
    import ctypes as C

    from bdd_amd import capi
    L = capi.lib()
    n, lds, thr = C.c_uint32(0), C.c_uint32(0), C.c_uint64(0)
    capi.check(L.bddmma_device_chip(0, C.byref(n), C.byref(lds), C.byref(thr)), None)
    prop = torch.cuda.get_device_properties(0)
    assert n.value == prop.multi_processor_count
    assert lds.value >= 64 * 1024 and thr.value == n.value * prop.max_threads_per_multi_processor
    assert L.bddmma_device_chip(L.bddmma_device_count(), None, None, None) != 0
    col, costs = random_set_cover(300, 200, 5, seed=2)
    s = bdd_hip_parallel_mma(col, costs, precision="float")
    held, alloc = s.device_bytes(), s.device_allocated_bytes()
    assert 0 < held <= alloc <= held + (8 << 20), (held, alloc)
    s.iterations(3)
    assert np.isfinite(s.lower_bound())
    s.close()
