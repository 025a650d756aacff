"""Host-side mirror of the reference solver classes for the `cuda parallel mma` path.

`bdd_hip_parallel_mma` has the public surface of `LPMP::bdd_cuda_parallel_mma<REAL>` +
`bdd_cuda_base<REAL>` (reference: include/bdd_solver/bdd_cuda_parallel_mma.h:7-52,
include/bdd_solver/bdd_cuda_base.h:58-229) — same method names, argument meaning and error
behaviour — so the parity tests read like test/test_cuda_parallel_mma.cu and
test/test_bdd_cuda_*.cpp.  All compute happens in the HIP library behind the C-ABI.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .bdd_collection import BddCollection


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _is_dev(x):
    """a device buffer: anything with data_ptr() living on a GPU (torch CUDA tensors) — the thrust::device_vector
    overloads of the reference.  Passed through the C-ABI as a raw pointer with on_device = 1."""
    return hasattr(x, "data_ptr") and bool(getattr(x, "is_cuda", False))


def _dev_ptr(x, n, dtype):
    """raw pointer of a contiguous device buffer holding >= n elements of `dtype`"""
    assert x.is_contiguous() and x.numel() >= n, "device buffer too small or not contiguous"
    assert x.element_size() == np.dtype(dtype).itemsize, "device buffer has the wrong element type"
    return C.c_void_p(x.data_ptr())


class bdd_hip_parallel_mma:
    """Drop-in for bdd_cuda_parallel_mma<REAL> (value_type = float | double)."""

    def __init__(self, bdd_col: BddCollection, costs_hi=None, precision: str = "double", device: int = 0,
                 pack_width: int = 0, wide_pack_width: int = 0, deterministic: bool = False,
                 vars_per_bin: int = 0, stage_cap: int = 0, waves_per_block: int = 0, keep_bdd_order: bool = False,
                 resident_sweeps: int = 0, exchange_by_variable: int = 0, variant_flags: int = 0, pack_fill: int = 0, pack_stagger: int = 0, _handle=None):
        self._L = capi.lib()
        self.value_type = {"double": np.float64, "float": np.float32, "single": np.float32}[precision]
        self._prec = capi.F64 if self.value_type == np.float64 else capi.F32
        if _handle is not None:
            self._h = _handle
            return
        instr = np.ascontiguousarray(bdd_col.instr, dtype=np.uint64)
        delims = np.ascontiguousarray(bdd_col.delims, dtype=np.uint64)
        opts = capi.Options(pack_width, wide_pack_width, 1 if deterministic else 0, vars_per_bin, stage_cap, waves_per_block)
        opts.keep_bdd_order = int(keep_bdd_order)   # 0 (default): BDDs of equal shape are packed together; 1: input order; 2: include/bdd_mma.h
        opts.resident_sweeps = int(resident_sweeps)         # 0 automatic, 1 off, 2 on
        opts.exchange_by_variable = int(exchange_by_variable)             # 2: entries by (variable, bdd)
        opts.pack_fill = int(pack_fill)
        opts.pack_stagger = int(pack_stagger)
        opts.variant_flags = int(variant_flags)             # bit 0 / 1: backward / forward narrow + wide sweeps as two launches
        h = C.c_void_p()
        costs = None if costs_hi is None else np.ascontiguousarray(costs_hi, dtype=np.float64)
        rc = self._L.bddmma_create(C.byref(h), self._prec, device, _ptr(instr), _ptr(delims), bdd_col.nr_bdds(),
                                   _ptr(costs), 0 if costs is None else costs.size, C.byref(opts))
        capi.check(rc, None)
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._L.bddmma_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        capi.check(rc, self._h)

    # ---- sizes (bdd_cuda_base.h:98-116)
    def nr_variables(self): return int(self._L.bddmma_nr_variables(self._h))
    def nr_layers(self): return int(self._L.bddmma_nr_layers(self._h))
    def nr_bdd_nodes(self): return int(self._L.bddmma_nr_bdd_nodes(self._h))
    def nr_hops(self): return int(self._L.bddmma_nr_hops(self._h))
    def nr_packs(self): return int(self._L.bddmma_nr_packs(self._h))

    SWEEP_KINDS = ("none", "mixed", "streaming1", "streaming2", "streaming3", "resident1", "resident2")

    def solve_sweep_kind(self) -> str:
        """which kernels run the narrow packs' solve sweeps (include/bdd_mma.h: BDDMMA_SWEEPS_*)"""
        return self.SWEEP_KINDS[int(self._L.bddmma_solve_sweep_kind(self._h))]
    def fused_small(self) -> bool:
        """whole iterations run inside one launch (the instance fits one workgroup; csrc/kernels/small.hpp)"""
        return bool(self._L.bddmma_fused_small(self._h))
    def nontemporal_loads(self) -> bool:
        """the solve sweeps run in the instantiation that loads potentials and staging tables non-temporally (footprint beyond the caches' reach)"""
        return bool(self._L.bddmma_nontemporal_loads(self._h))
    def device_bytes(self): return int(self._L.bddmma_device_bytes(self._h))
    def device_allocated_bytes(self): return int(self._L.bddmma_device_allocated_bytes(self._h))

    def nr_bdds(self, var=None):
        if var is None:
            return int(self._L.bddmma_nr_bdds(self._h))
        return int(self.get_num_bdds_per_var()[var])

    def get_num_bdds_per_var(self):
        out = np.zeros(self.nr_variables(), np.int32)
        self._ck(self._L.bddmma_num_bdds_per_var(self._h, _ptr(out)))
        return out

    def get_primal_variable_index(self):
        out = np.zeros(self.nr_layers(), np.int32)
        self._ck(self._L.bddmma_layer_variables(self._h, _ptr(out)))
        return out

    def get_bdd_index(self):
        out = np.zeros(self.nr_layers(), np.int32)
        self._ck(self._L.bddmma_layer_bdds(self._h, _ptr(out)))
        return out

    def nodes_per_hop(self):
        out = np.zeros(self.nr_hops(), np.uint64)
        self._ck(self._L.bddmma_nodes_per_hop(self._h, _ptr(out)))
        return out

    def layers_per_hop(self):
        out = np.zeros(self.nr_hops(), np.uint64)
        self._ck(self._L.bddmma_layers_per_hop(self._h, _ptr(out)))
        return out

    def bdd_major_order(self):
        """Permutation taking internal layer order to BDD-major order (bdd ascending, hop ascending):
        the layer order of the reference CPU solver (bdd_parallel_mma_base.cpp:75-170)."""
        return np.argsort(self.get_bdd_index(), kind="stable")

    # ---- costs
    def update_costs(self, cost_delta_0, cost_delta_1):
        if _is_dev(cost_delta_0) or _is_dev(cost_delta_1):   # update_costs(device_vector<REAL>, device_vector<REAL>), bdd_cuda_base.cu:476-500
            n0 = cost_delta_0.numel() if cost_delta_0 is not None else 0
            n1 = cost_delta_1.numel() if cost_delta_1 is not None else 0
            self._ck(self._L.bddmma_update_costs(self._h, _dev_ptr(cost_delta_0, n0, self.value_type) if n0 else None, n0,
                                                 _dev_ptr(cost_delta_1, n1, self.value_type) if n1 else None, n1, self._prec, 1))
            return
        lo = np.ascontiguousarray(cost_delta_0, dtype=np.float64)
        hi = np.ascontiguousarray(cost_delta_1, dtype=np.float64)
        self._ck(self._L.bddmma_update_costs(self._h, _ptr(lo), lo.size, _ptr(hi), hi.size, capi.F64, 0))

    def set_cost(self, c, var):
        self._ck(self._L.bddmma_set_cost(self._h, float(c), int(var)))

    def get_solver_costs(self, out=None):
        n = self.nr_layers()
        if out is not None:   # three device buffers
            self._ck(self._L.bddmma_get_solver_costs(self._h, *(_dev_ptr(x, n, self.value_type) for x in out), 1))
            return out
        lo, hi, mm = (np.zeros(n, self.value_type) for _ in range(3))
        self._ck(self._L.bddmma_get_solver_costs(self._h, _ptr(lo), _ptr(hi), _ptr(mm), 0))
        return lo, hi, mm

    def set_solver_costs(self, lo, hi, mm):
        if _is_dev(lo):
            n = self.nr_layers()
            self._ck(self._L.bddmma_set_solver_costs(self._h, *(_dev_ptr(x, n, self.value_type) for x in (lo, hi, mm)), 1))
            return
        lo, hi, mm = (np.ascontiguousarray(x, dtype=self.value_type) for x in (lo, hi, mm))
        self._ck(self._L.bddmma_set_solver_costs(self._h, _ptr(lo), _ptr(hi), _ptr(mm), 0))

    def get_primal_objective_vector(self, out):
        """compute_primal_objective_vec into a device buffer (bdd_cuda_base.cu:1352-1362)"""
        self._ck(self._L.bddmma_primal_objective_vec(self._h, _dev_ptr(out, self.nr_variables(), self.value_type), 1))
        return out

    def get_primal_objective_vector_host(self):
        out = np.zeros(self.nr_variables(), self.value_type)
        self._ck(self._L.bddmma_primal_objective_vec(self._h, _ptr(out), 0))
        return out

    # ---- sweeps
    def forward_run(self): self._ck(self._L.bddmma_forward_run(self._h))
    def backward_run(self): self._ck(self._L.bddmma_backward_run(self._h))

    def lower_bound(self) -> float:
        lb = C.c_double()
        self._ck(self._L.bddmma_lower_bound(self._h, C.byref(lb)))
        return lb.value

    def lower_bound_per_bdd(self, out=None):
        if out is not None:
            self._ck(self._L.bddmma_lower_bound_per_bdd(self._h, _dev_ptr(out, self.nr_bdds(), self.value_type), 1))
            return out
        out = np.zeros(self.nr_bdds(), self.value_type)
        self._ck(self._L.bddmma_lower_bound_per_bdd(self._h, _ptr(out), 0))
        return out

    # ---- parallel mma
    def iteration(self, omega=0.5):
        self._ck(self._L.bddmma_iteration(self._h, float(omega)))

    def iterations(self, n, omega=0.5):
        self._ck(self._L.bddmma_iterations(self._h, float(omega), int(n)))

    def forward_mm(self, omega, delta_lo_hi):
        if _is_dev(delta_lo_hi):   # forward_mm(omega, device_vector<REAL>&), bdd_cuda_parallel_mma.cu:207-257
            self._ck(self._L.bddmma_forward_mm(self._h, float(omega), _dev_ptr(delta_lo_hi, 2 * self.nr_variables(), self.value_type), 1))
            return
        assert delta_lo_hi.dtype == self.value_type and delta_lo_hi.size == 2 * self.nr_variables()
        self._ck(self._L.bddmma_forward_mm(self._h, float(omega), _ptr(delta_lo_hi), 0))

    def backward_mm(self, omega, delta_lo_hi):
        if _is_dev(delta_lo_hi):
            self._ck(self._L.bddmma_backward_mm(self._h, float(omega), _dev_ptr(delta_lo_hi, 2 * self.nr_variables(), self.value_type), 1))
            return
        assert delta_lo_hi.dtype == self.value_type and delta_lo_hi.size == 2 * self.nr_variables()
        self._ck(self._L.bddmma_backward_mm(self._h, float(omega), _ptr(delta_lo_hi), 0))

    def normalize_delta(self, delta_lo_hi):
        if _is_dev(delta_lo_hi):
            self._ck(self._L.bddmma_normalize_delta(self._h, _dev_ptr(delta_lo_hi, 2 * self.nr_variables(), self.value_type), 1))
            return
        self._ck(self._L.bddmma_normalize_delta(self._h, _ptr(delta_lo_hi), 0))

    def distribute_delta(self):
        self._ck(self._L.bddmma_distribute_delta(self._h))

    def set_delta(self, delta_lo_hi):
        if _is_dev(delta_lo_hi):
            self._ck(self._L.bddmma_set_delta(self._h, _dev_ptr(delta_lo_hi, 2 * self.nr_variables(), self.value_type), 1))
            return
        d = np.ascontiguousarray(delta_lo_hi, dtype=self.value_type)
        self._ck(self._L.bddmma_set_delta(self._h, _ptr(d), 0))

    def get_delta(self, out=None):
        if out is not None:
            self._ck(self._L.bddmma_get_delta(self._h, _dev_ptr(out, 2 * self.nr_variables(), self.value_type), 1))
            return out
        out = np.zeros(2 * self.nr_variables(), self.value_type)
        self._ck(self._L.bddmma_get_delta(self._h, _ptr(out), 0))
        return out

    # ---- min-marginals / solutions
    def min_marginals_cuda(self, get_sorted=True, out=None):
        n = self.nr_layers()
        if out is not None:   # (int32 var, REAL mm0, REAL mm1) device buffers, as min_marginals_cuda returns them (bdd_cuda_base.cu:716-749)
            v, m0, m1 = out
            self._ck(self._L.bddmma_min_marginals(self._h, 1 if get_sorted else 0, _dev_ptr(v, n, np.int32), _dev_ptr(m0, n, self.value_type),
                                                  _dev_ptr(m1, n, self.value_type), 1))
            return out
        var = np.zeros(n, np.int32)
        mm0, mm1 = np.zeros(n, self.value_type), np.zeros(n, self.value_type)
        self._ck(self._L.bddmma_min_marginals(self._h, 1 if get_sorted else 0, _ptr(var), _ptr(mm0), _ptr(mm1), 0))
        return var, mm0, mm1

    def min_marginals(self):
        """two_dim_variable_array<array<double,2>>[var][bdd] (bdd_cuda_base.cu:751-786) as a list of (k,2) arrays."""
        var, mm0, mm1 = self.min_marginals_cuda(True)
        nb = self.get_num_bdds_per_var()
        ptr = np.concatenate([[0], np.cumsum(nb)])
        return [np.stack([mm0[ptr[v]:ptr[v + 1]], mm1[ptr[v]:ptr[v + 1]]], axis=1).astype(np.float64)
                for v in range(self.nr_variables())]

    def min_marginal_diff(self, out=None):
        """mm1 - mm0 per layer (compute_and_set_min_marginal_diff, bdd_cuda_parallel_mma_py.cu:56-72); `out`: device buffer"""
        if out is not None:
            self._ck(self._L.bddmma_min_marginal_diff(self._h, _dev_ptr(out, self.nr_layers(), self.value_type), 1))
            return out
        res = np.zeros(self.nr_layers(), self.value_type)
        self._ck(self._L.bddmma_min_marginal_diff(self._h, _ptr(res), 0))
        return res

    def bdds_solution_vec(self, out=None):
        if out is not None:   # device_vector<char> (bdd_cuda_base.cu:1138-1145)
            self._ck(self._L.bddmma_bdds_solution(self._h, 0, _dev_ptr(out, self.nr_layers(), np.int8), 1))
            return out
        out = np.zeros(self.nr_layers(), np.int8)
        self._ck(self._L.bddmma_bdds_solution(self._h, 0, _ptr(out), 0))
        return out

    def bdds_solution(self):
        out = np.zeros(self.nr_layers(), np.int8)
        self._ck(self._L.bddmma_bdds_solution(self._h, 1, _ptr(out), 0))
        nb = self.get_num_bdds_per_var()
        ptr = np.concatenate([[0], np.cumsum(nb)])
        return [out[ptr[v]:ptr[v + 1]].astype(np.float64) for v in range(self.nr_variables())]

    # ---- L-BFGS support
    def net_solver_costs(self, out=None):
        if out is not None:
            self._ck(self._L.bddmma_net_solver_costs(self._h, _dev_ptr(out, self.nr_layers(), self.value_type), 1))
            return out
        out = np.zeros(self.nr_layers(), self.value_type)
        self._ck(self._L.bddmma_net_solver_costs(self._h, _ptr(out), 0))
        return out

    def make_dual_feasible(self, d):
        if _is_dev(d):
            self._ck(self._L.bddmma_make_dual_feasible(self._h, _dev_ptr(d, self.nr_layers(), self.value_type), 1))
            return
        assert d.dtype == self.value_type and d.size == self.nr_layers()
        self._ck(self._L.bddmma_make_dual_feasible(self._h, _ptr(d), 0))

    def gradient_step(self, g, step_size):
        if _is_dev(g):
            self._ck(self._L.bddmma_gradient_step(self._h, _dev_ptr(g, self.nr_layers(), self.value_type), float(step_size), 1))
            return
        g = np.ascontiguousarray(g, dtype=self.value_type)
        self._ck(self._L.bddmma_gradient_step(self._h, _ptr(g), float(step_size), 0))

    # ---- primal rounding
    def perturb_primal_costs(self, cur_delta, round_index=0, seed=0, lbfgs=None):
        """one round of perturb_primal_costs (incremental_mm_agreement_rounding_cuda.cu:262-331)
        -> dict(counts = (#one, #zero, #equal, #inconsistent), sol, cost_delta_0, cost_delta_1)"""
        n = self.nr_variables()
        counts = (C.c_uint32 * 4)()
        sol = np.zeros(n, np.int8)
        c0, c1 = np.zeros(n, self.value_type), np.zeros(n, self.value_type)
        self._ck(self._L.bddmma_perturb_primal_costs(self._h, lbfgs._h if lbfgs is not None else None, float(cur_delta), int(round_index),
                                                     int(seed), counts, _ptr(sol), _ptr(c0), _ptr(c1)))
        return dict(counts=tuple(int(c) for c in counts), sol=sol, cost_delta_0=c0, cost_delta_1=c1)

    # ---- checkpoint (bdd_cuda_base.cu:1486-1550; pickle in bdd_cuda_parallel_mma_py.cu:15-38)
    def save(self, path: str):
        self._ck(self._L.bddmma_save(self._h, path.encode()))

    @classmethod
    def load(cls, path: str, device: int = 0):
        L = capi.lib()
        h = C.c_void_p()
        capi.check(L.bddmma_load(C.byref(h), device, path.encode()), None)
        prec = "double" if L.bddmma_precision(h) == capi.F64 else "float"
        return cls(None, precision=prec, _handle=h)

    # ---- measurement
    def synchronize(self): self._ck(self._L.bddmma_synchronize(self._h))
    def set_profiling(self, on, stride: int = 1):
        """record hipEvent pairs around the launches of every `stride`-th iteration (0/False: off)"""
        self._ck(self._L.bddmma_set_profiling(self._h, int(stride) if on else 0))

    def get_profile(self):
        p = capi.Profile()
        self._ck(self._L.bddmma_get_profile(self._h, C.byref(p)))
        return {"launches": list(p.launches), "total_ms": list(p.total_ms)}

    def time_kernel(self, kind: int, reps: int) -> float:
        """average ms per launch of one kernel class (see bddmma_time_kernel)"""
        ms = C.c_double()
        self._ck(self._L.bddmma_time_kernel(self._h, int(kind), int(reps), C.byref(ms)))
        return ms.value / reps

    def time_iterations(self, n, omega=0.5) -> float:
        ms = C.c_double()
        self._ck(self._L.bddmma_time_iterations(self._h, float(omega), int(n), C.byref(ms)))
        return ms.value


class bdd_hip_lbfgs:
    """lbfgs<bdd_cuda_parallel_mma<REAL>, ...> (include/bdd_solver/lbfgs.h:35-111) over a HIP solver."""

    def __init__(self, solver: bdd_hip_parallel_mma, history_size=5, init_step_size=1e-6, req_rel_lb_increase=1e-6,
                 step_size_decrease_factor=0.8, step_size_increase_factor=1.1):
        self.solver = solver
        self._L = capi.lib()
        p = capi.LbfgsParams(history_size, init_step_size, req_rel_lb_increase, step_size_decrease_factor,
                             step_size_increase_factor)
        h = C.c_void_p()
        capi.check(self._L.bddmma_lbfgs_create(C.byref(h), solver._h, C.byref(p)), solver._h)
        self._h = h

    def close(self):
        """release the history buffers; call before closing the wrapped solver"""
        if getattr(self, "_h", None):
            self._L.bddmma_lbfgs_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def iteration(self):
        capi.check(self._L.bddmma_lbfgs_iteration(self._h), self.solver._h)

    def lower_bound(self):
        return self.solver.lower_bound()

    def flush(self):
        capi.check(self._L.bddmma_lbfgs_flush(self._h), self.solver._h)

    def state(self):
        st = capi.LbfgsState()
        capi.check(self._L.bddmma_lbfgs_get_state(self._h, C.byref(st)), self.solver._h)
        return {k: getattr(st, k) for k, _ in capi.LbfgsState._fields_}

    def update_costs(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        capi.check(self._L.bddmma_lbfgs_update_costs(self._h, _ptr(lo), lo.size, _ptr(hi), hi.size, capi.F64, 0),
                   self.solver._h)


def run_solver(solver, max_iter=1000, tolerance=1e-6, improvement_slope=1e-9, time_limit=3600.0, verbose=False,
               lbfgs: bdd_hip_lbfgs = None, host_loop: bool = False):
    """run_solver<SOLVER>() of include/run_solver_util.h:10-77 (executed inside the library).  host_loop: the reference's literal loop
    (a host round trip for the bound every iteration) instead of the device-resident one; same result."""
    L = capi.lib()
    res = capi.RunResult()
    base = solver.solver if isinstance(solver, bdd_hip_lbfgs) else solver
    lb = solver if isinstance(solver, bdd_hip_lbfgs) else lbfgs
    capi.check((L.bddmma_run_solver_host_loop if host_loop else L.bddmma_run_solver)(base._h, lb._h if lb else None, int(max_iter), float(tolerance),
                                   float(improvement_slope), float(time_limit), 1 if verbose else 0, C.byref(res)), base._h)
    return dict(iterations=int(res.iterations), lb_initial=res.lb_initial, lb_final=res.lb_final,
                seconds=res.seconds, stop_reason=int(res.stop_reason))
