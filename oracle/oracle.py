"""ctypes bindings of the CPU oracle (oracle/libmma_oracle.so) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY — see oracle/mma_oracle.c.  `Oracle` restates the reference's CPU
`parallel mma` solver; `RefCollection` / `RefMma` call the reference's own compiled code
(oracle/_ref/libref_driver.so, built by oracle/Makefile from /root/reference).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None


def build(verbose: bool = False) -> None:
    """make -C oracle (the C restatement; oracle/_ref too when /root/reference exists)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("building the oracle failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmma_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def ref_available() -> bool:
    return os.path.exists(os.path.join(_HERE, "_ref", "libref_driver.so"))


def ref():
    global _REF
    if _REF is None:
        _REF = C.CDLL(os.path.join(_HERE, "_ref", "libref_driver.so"))
    return _REF


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """CPU restatement of bdd_parallel_mma_base<bdd_branch_instruction<REAL,uint16_t>>."""

    def __init__(self, col, costs_hi=None, precision: str = "double", threads: int = 1):
        self.suf = {"double": "_f64", "float": "_f32"}[precision]
        self.dtype = np.float64 if precision == "double" else np.float32
        self.L = lib()
        instr = np.ascontiguousarray(col.instr, dtype=np.uint64)
        delims = np.ascontiguousarray(col.delims, dtype=np.uint64)
        f = self._f("oracle_create")
        f.restype = C.c_void_p
        self.h = C.c_void_p(f(_p(instr), _p(delims), C.c_uint64(col.nr_bdds())))
        self._f("oracle_set_num_threads")(self.h, C.c_int(threads))
        if costs_hi is not None:
            self.update_costs([], costs_hi)

    def _f(self, name):
        return getattr(self.L, name + self.suf)

    def __del__(self):
        try:
            self._f("oracle_destroy")(self.h)
        except Exception:
            pass

    def set_threads(self, n: int):
        self._f("oracle_set_num_threads")(self.h, C.c_int(n))

    def _u64(self, name):
        f = self._f(name)
        f.restype = C.c_uint64
        return int(f(self.h))

    def nr_variables(self): return self._u64("oracle_nr_variables")
    def nr_bdds(self): return self._u64("oracle_nr_bdds")
    def nr_layers(self): return self._u64("oracle_nr_layers")

    def nr_bdds_per_var(self):
        out = np.zeros(self.nr_variables(), np.int64)
        self._f("oracle_nr_bdds_per_var")(self.h, _p(out))
        return out

    def layer_info(self):
        var = np.zeros(self.nr_layers(), np.int64)
        bdd = np.zeros(self.nr_layers(), np.int64)
        self._f("oracle_layer_info")(self.h, _p(var), _p(bdd))
        return var, bdd

    def update_costs(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        self._f("oracle_update_costs")(self.h, _p(lo), C.c_uint64(lo.size), _p(hi), C.c_uint64(hi.size))

    def get_costs(self):
        lo = np.zeros(self.nr_layers(), self.dtype)
        hi = np.zeros(self.nr_layers(), self.dtype)
        self._f("oracle_get_costs")(self.h, _p(lo), _p(hi))
        return lo, hi

    def backward_run(self): self._f("oracle_backward_run")(self.h)
    def forward_run(self): self._f("oracle_forward_run")(self.h)

    def lower_bound(self) -> float:
        f = self._f("oracle_lower_bound")
        f.restype = C.c_double
        return float(f(self.h))

    def lower_bound_per_bdd(self):
        out = np.zeros(self.nr_bdds(), self.dtype)
        self._f("oracle_lower_bound_per_bdd")(self.h, _p(out))
        return out

    def _real(self, x):
        return C.c_double(x) if self.dtype == np.float64 else C.c_float(x)

    def forward_mm(self, omega, delta):
        assert delta.dtype == self.dtype and delta.size == 2 * self.nr_variables()
        self._f("oracle_forward_mm")(self.h, self._real(omega), _p(delta))

    def backward_mm(self, omega, delta) -> float:
        assert delta.dtype == self.dtype and delta.size == 2 * self.nr_variables()
        f = self._f("oracle_backward_mm")
        f.restype = C.c_double
        return float(f(self.h, self._real(omega), _p(delta)))

    def iteration(self): self._f("oracle_iteration")(self.h)
    def distribute_delta(self): self._f("oracle_distribute_delta")(self.h)

    def delta_in(self):
        out = np.zeros(2 * self.nr_variables(), self.dtype)
        self._f("oracle_get_delta_in")(self.h, _p(out))
        return out

    def min_marginals(self):
        """(n_layers, 2) doubles in BDD-major layer order."""
        out = np.zeros((self.nr_layers(), 2), np.float64)
        self._f("oracle_min_marginals")(self.h, _p(out))
        return out

    def bdds_solution_vec(self):
        out = np.zeros(self.nr_layers(), np.int8)
        self._f("oracle_bdds_solution_vec")(self.h, _p(out))
        return out

    def net_solver_costs(self):
        """hi - lo + the layer's own deferred min-marginal difference (GPU definition, bdd_cuda_parallel_mma.cu:432-463)."""
        out = np.zeros(self.nr_layers(), self.dtype)
        self._f("oracle_net_solver_costs")(self.h, _p(out))
        return out

    def mm_last(self):
        out = np.zeros(self.nr_layers(), self.dtype)
        self._f("oracle_get_mm_last")(self.h, _p(out))
        return out

    def make_dual_feasible(self, d):
        assert d.dtype == self.dtype
        self._f("oracle_make_dual_feasible")(self.h, _p(d))

    def gradient_step(self, d, step):
        assert d.dtype == self.dtype
        self._f("oracle_gradient_step")(self.h, _p(d), C.c_double(step))


class CudaRuleOracle:
    """CPU restatement of the reference's GPU solver bdd_cuda_parallel_mma<REAL> (oracle/cuda_rule_oracle.c): omega is an
    argument, a layer with a non-finite min-marginal gets no update.  Layers are in BDD-major order, as in `Oracle`."""

    def __init__(self, col, costs_hi=None, precision: str = "double"):
        self.suf = {"double": "_f64", "float": "_f32"}[precision]
        self.dtype = np.float64 if precision == "double" else np.float32
        self.L = lib()
        instr = np.ascontiguousarray(col.instr, dtype=np.uint64)
        delims = np.ascontiguousarray(col.delims, dtype=np.uint64)
        f = self._f("cr_create")
        f.restype = C.c_void_p
        self.h = C.c_void_p(f(_p(instr), _p(delims), C.c_uint64(col.nr_bdds())))
        if costs_hi is not None:
            self.update_costs([], costs_hi)

    def _f(self, name):
        return getattr(self.L, name + self.suf)

    def __del__(self):
        try:
            self._f("cr_destroy")(self.h)
        except Exception:
            pass

    def _u64(self, name):
        f = self._f(name)
        f.restype = C.c_uint64
        return int(f(self.h))

    def nr_variables(self): return self._u64("cr_nr_variables")
    def nr_bdds(self): return self._u64("cr_nr_bdds")
    def nr_layers(self): return self._u64("cr_nr_layers")

    def _real(self, x):
        return C.c_double(x) if self.dtype == np.float64 else C.c_float(x)

    def layer_info(self):
        var = np.zeros(self.nr_layers(), np.int64)
        bdd = np.zeros(self.nr_layers(), np.int64)
        self._f("cr_layer_info")(self.h, _p(var), _p(bdd))
        return var, bdd

    def update_costs(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        self._f("cr_update_costs")(self.h, _p(lo), C.c_uint64(lo.size), _p(hi), C.c_uint64(hi.size))

    def get_costs(self):
        lo = np.zeros(self.nr_layers(), self.dtype)
        hi = np.zeros(self.nr_layers(), self.dtype)
        self._f("cr_get_layer_costs")(self.h, _p(lo), _p(hi))
        return lo, hi

    def set_costs(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=self.dtype)
        hi = np.ascontiguousarray(hi, dtype=self.dtype)
        assert lo.size == hi.size == self.nr_layers()
        self._f("cr_set_layer_costs")(self.h, _p(lo), _p(hi))

    def backward_run(self): self._f("cr_backward_run")(self.h)
    def forward_run(self): self._f("cr_forward_run")(self.h)

    def lower_bound(self) -> float:
        f = self._f("cr_lower_bound")
        f.restype = C.c_double
        return float(f(self.h))

    def forward_mm(self, omega, delta):
        assert delta.dtype == self.dtype and delta.size == 2 * self.nr_variables()
        self._f("cr_forward_mm")(self.h, self._real(omega), _p(delta))

    def backward_mm(self, omega, delta):
        assert delta.dtype == self.dtype and delta.size == 2 * self.nr_variables()
        f = self._f("cr_backward_mm")
        f.restype = C.c_int
        if f(self.h, self._real(omega), _p(delta)) != 0:
            raise RuntimeError("backward_mm needs a valid forward state")

    def normalize_delta(self, delta):
        assert delta.dtype == self.dtype and delta.size == 2 * self.nr_variables()
        self._f("cr_normalize_delta")(self.h, _p(delta))

    def iteration(self, omega=0.5): self._f("cr_iteration")(self.h, self._real(omega))
    def distribute_delta(self): self._f("cr_distribute_delta")(self.h)

    def delta(self):
        out = np.zeros(2 * self.nr_variables(), self.dtype)
        self._f("cr_get_delta")(self.h, _p(out))
        return out

    def mm(self):
        """deferred min-marginal differences (deffered_mm_diff_) of the last pass, per layer"""
        out = np.zeros(self.nr_layers(), self.dtype)
        self._f("cr_get_mm")(self.h, _p(out))
        return out

    def min_marginals(self):
        mm0 = np.zeros(self.nr_layers(), self.dtype)
        mm1 = np.zeros(self.nr_layers(), self.dtype)
        self._f("cr_min_marginals")(self.h, _p(mm0), _p(mm1))
        return mm0, mm1


class RefCollection:
    """BDD::bdd_collection of the reference (compiled from /root/reference), via oracle/_ref."""

    def __init__(self):
        self.R = ref()
        self.R.ref_col_new.restype = C.c_void_p
        self.h = C.c_void_p(self.R.ref_col_new())

    def __del__(self):
        try:
            self.R.ref_col_free(self.h)
        except Exception:
            pass

    def _add(self, fn, variables, *extra):
        v = np.ascontiguousarray(variables, dtype=np.uint64)
        fn.restype = C.c_long
        return int(fn(self.h, C.c_size_t(v.size), *extra, _p(v)))

    def add_simplex(self, variables): return self._add(self.R.ref_col_add_simplex, variables)
    def add_covering(self, variables): return self._add(self.R.ref_col_add_covering, variables)
    def add_all_equal(self, variables): return self._add(self.R.ref_col_add_all_equal, variables)
    def add_cardinality(self, variables, k): return self._add(self.R.ref_col_add_cardinality, variables, C.c_size_t(k))

    def add_linear(self, coeffs, ineq, rhs, variables):
        v = np.ascontiguousarray(variables, dtype=np.uint64)
        c = np.ascontiguousarray(coeffs, dtype=np.int32)
        code = {"<=": -1, "=": 0, ">=": 1}[ineq]
        self.R.ref_col_add_linear.restype = C.c_long
        return int(self.R.ref_col_add_linear(self.h, C.c_size_t(v.size), _p(c), C.c_int(code), C.c_int(rhs), _p(v)))

    def split_qbdd(self, bdd_nr, chunk_size, aux_var_start, with_implication_bdd=False):
        """reference split_qbdd + removal of the original BDD; -> (#BDDs produced, next free aux variable)"""
        na = C.c_size_t(0)
        self.R.ref_col_split_qbdd.restype = C.c_long
        n = self.R.ref_col_split_qbdd(self.h, C.c_size_t(bdd_nr), C.c_size_t(chunk_size), C.c_size_t(aux_var_start),
                                      C.c_int(int(with_implication_bdd)), C.byref(na))
        return int(n), int(na.value)

    def nr_bdds(self):
        self.R.ref_col_nr_bdds.restype = C.c_size_t
        return int(self.R.ref_col_nr_bdds(self.h))

    def _text(self, fn, *args):
        fn.restype = C.c_size_t
        n = int(fn(self.h, *args, None, C.c_size_t(0)))
        buf = C.create_string_buffer(n + 1)
        fn(self.h, *args, buf, C.c_size_t(n + 1))
        return buf.value.decode()

    def write_bdd_lp(self, costs):
        """reference bdd_collection::write_bdd_lp (bdd_collection.h:731-830)"""
        c = np.ascontiguousarray(costs, dtype=np.float64)
        return self._text(self.R.ref_col_write_bdd_lp, _p(c), C.c_size_t(c.size))

    def export_graphviz(self, bdd_nr):
        """reference bdd_collection::export_graphviz (bdd_collection.h:663-729)"""
        return self._text(self.R.ref_col_export_graphviz, C.c_size_t(bdd_nr))

    def export(self):
        """-> bdd_amd.BddCollection holding the reference's instructions."""
        from bdd_amd.bdd_collection import BddCollection
        self.R.ref_col_nr_instructions.restype = C.c_size_t
        n = int(self.R.ref_col_nr_instructions(self.h))
        instr = np.zeros((n, 3), np.uint64)
        delims = np.zeros(self.nr_bdds() + 1, np.uint64)
        self.R.ref_col_export(self.h, _p(instr), _p(delims))
        return BddCollection.from_arrays(instr, delims)

    def flatten_dropin(self):
        """(instructions [n, 3], delimiters) as LPMP::bdd_hip_parallel_mma<REAL>::flatten — the drop-in class's template constructor
        instantiated with the reference's BDD::bdd_collection — produces them (bdd_amd/csrc/bdd_hip_parallel_mma.hpp)."""
        self.R.ref_col_nr_instructions.restype = C.c_size_t
        self.R.ref_col_flatten_dropin.restype = C.c_size_t
        n = int(self.R.ref_col_nr_instructions(self.h))
        instr = np.zeros((n, 3), np.uint64)
        delims = np.zeros(self.nr_bdds() + 1, np.uint64)
        got = int(self.R.ref_col_flatten_dropin(self.h, _p(instr), _p(delims)))
        assert got == n, (got, n)
        return instr, delims


class RefMma:
    """MMA over the reference's bdd_branch_instruction<REAL,uint16_t> node code (oracle/ref_driver.cpp)."""

    def __init__(self, refcol: RefCollection, precision="double"):
        self.R = ref()
        self.suf = {"double": "_f64", "float": "_f32"}[precision]
        self.dtype = np.float64 if precision == "double" else np.float32
        f = self._f("ref_mma_new")
        f.restype = C.c_void_p
        self.h = C.c_void_p(f(refcol.h))

    def _f(self, name):
        return getattr(self.R, name + self.suf)

    def __del__(self):
        try:
            self._f("ref_mma_free")(self.h)
        except Exception:
            pass

    def nr_variables(self):
        f = self._f("ref_mma_nr_variables")
        f.restype = C.c_size_t
        return int(f(self.h))

    def update_costs(self, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.float64)
        hi = np.ascontiguousarray(hi, dtype=np.float64)
        self._f("ref_mma_update_costs")(self.h, _p(lo), C.c_size_t(lo.size), _p(hi), C.c_size_t(hi.size))

    def lower_bound(self):
        f = self._f("ref_mma_lower_bound")
        f.restype = C.c_double
        return float(f(self.h))

    def iteration(self):
        f = self._f("ref_mma_iteration")
        f.restype = C.c_double
        return float(f(self.h))

    def _real(self, x):
        return C.c_double(x) if self.dtype == np.float64 else C.c_float(x)

    def forward_mm(self, omega, delta):
        self._f("ref_mma_forward_mm")(self.h, self._real(omega), _p(delta))

    def backward_mm(self, omega, delta):
        f = self._f("ref_mma_backward_mm")
        f.restype = C.c_double
        return float(f(self.h, self._real(omega), _p(delta)))
