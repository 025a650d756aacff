"""ctypes binding of the C-ABI library (include/bdd_mma.h).

This is the binding a maintainer of a Python front-end (the reference's `bdd_solver_py`,
src/bdd_solver/bdd_solver_py.cpp:9-20) would add.  It loads the in-tree
bdd_amd/csrc/libbdd_mma_hip.so and fails loudly if the library is missing: there is no
CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
# BDDMMA_LIB: load another build of the same library (kernel experiments under build/); default: the in-tree build
LIB_PATH = os.path.abspath(os.environ["BDDMMA_LIB"]) if os.environ.get("BDDMMA_LIB") else os.path.join(_HERE, "csrc", "libbdd_mma_hip.so")

OK = 0
F32, F64 = 0, 1
K_FORWARD_MM, K_BACKWARD_MM, K_FINISH_DELTA, K_OTHER, K_COUNT = 0, 1, 2, 3, 4


class Options(C.Structure):
    _fields_ = [("pack_width", C.c_uint32), ("wide_pack_width", C.c_uint32), ("deterministic", C.c_uint32),
                ("vars_per_bin", C.c_uint32), ("stage_cap", C.c_uint32), ("waves_per_block", C.c_uint32),
                ("keep_bdd_order", C.c_uint32), ("resident_sweeps", C.c_uint32), ("exchange_by_variable", C.c_uint32),
                ("variant_flags", C.c_uint32), ("pack_fill", C.c_uint32), ("pack_stagger", C.c_uint32)]


class LbfgsParams(C.Structure):
    _fields_ = [("history_size", C.c_int32), ("init_step_size", C.c_double), ("req_rel_lb_increase", C.c_double),
                ("step_size_decrease_factor", C.c_double), ("step_size_increase_factor", C.c_double)]


class LbfgsState(C.Structure):
    _fields_ = [("step_size", C.c_double), ("last_applied_step", C.c_double), ("mma_iterations", C.c_uint64),
                ("lbfgs_iterations", C.c_uint64), ("history_entries", C.c_int32), ("num_unsuccessful_updates", C.c_int32),
                ("last_kind", C.c_int32), ("last_trials", C.c_int32)]


class RunResult(C.Structure):
    _fields_ = [("iterations", C.c_uint64), ("lb_initial", C.c_double), ("lb_final", C.c_double),
                ("seconds", C.c_double), ("stop_reason", C.c_int32)]


class Profile(C.Structure):
    _fields_ = [("launches", C.c_uint64 * K_COUNT), ("total_ms", C.c_double * K_COUNT)]


# every symbol include/bdd_mma.h declares: name -> (restype, argtypes)
_V, _U64, _I, _D = C.c_void_p, C.c_uint64, C.c_int, C.c_double
SIGNATURES = {
    "bddmma_create": (_I, [C.POINTER(_V), _I, _I, _V, _V, _U64, _V, _U64, C.POINTER(Options)]),
    "bddmma_destroy": (None, [_V]),
    "bddmma_device_count": (_I, []),
    "bddmma_device_chip": (_I, [_I, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    "bddmma_set_layout_threads": (_I, [_I]),
    "bddmma_set_thread_layout_threads": (_I, [_I]),
    "bddmma_last_error": (C.c_char_p, [_V]),
    "bddmma_nr_variables": (_U64, [_V]),
    "bddmma_nr_bdds": (_U64, [_V]),
    "bddmma_nr_layers": (_U64, [_V]),
    "bddmma_nr_bdd_nodes": (_U64, [_V]),
    "bddmma_nr_hops": (_U64, [_V]),
    "bddmma_nr_packs": (_U64, [_V]),
    "bddmma_solve_sweep_kind": (_I, [_V]),
    "bddmma_fused_small": (_I, [_V]),
    "bddmma_nontemporal_loads": (_I, [_V]),
    "bddmma_precision": (_I, [_V]),
    "bddmma_device": (_I, [_V]),
    "bddmma_num_bdds_per_var": (_I, [_V, _V]),
    "bddmma_layer_variables": (_I, [_V, _V]),
    "bddmma_layer_bdds": (_I, [_V, _V]),
    "bddmma_nodes_per_hop": (_I, [_V, _V]),
    "bddmma_layers_per_hop": (_I, [_V, _V]),
    "bddmma_update_costs": (_I, [_V, _V, _U64, _V, _U64, _I, _I]),
    "bddmma_set_cost": (_I, [_V, _D, _U64]),
    "bddmma_get_solver_costs": (_I, [_V, _V, _V, _V, _I]),
    "bddmma_set_solver_costs": (_I, [_V, _V, _V, _V, _I]),
    "bddmma_primal_objective_vec": (_I, [_V, _V, _I]),
    "bddmma_forward_run": (_I, [_V]),
    "bddmma_backward_run": (_I, [_V]),
    "bddmma_lower_bound": (_I, [_V, C.POINTER(_D)]),
    "bddmma_lower_bound_per_bdd": (_I, [_V, _V, _I]),
    "bddmma_iteration": (_I, [_V, _D]),
    "bddmma_iterations": (_I, [_V, _D, _U64]),
    "bddmma_forward_mm": (_I, [_V, _D, _V, _I]),
    "bddmma_backward_mm": (_I, [_V, _D, _V, _I]),
    "bddmma_normalize_delta": (_I, [_V, _V, _I]),
    "bddmma_distribute_delta": (_I, [_V]),
    "bddmma_get_delta": (_I, [_V, _V, _I]),
    "bddmma_set_delta": (_I, [_V, _V, _I]),
    "bddmma_min_marginals": (_I, [_V, _I, _V, _V, _V, _I]),
    "bddmma_min_marginal_diff": (_I, [_V, _V, _I]),
    "bddmma_bdds_solution": (_I, [_V, _I, _V, _I]),
    "bddmma_net_solver_costs": (_I, [_V, _V, _I]),
    "bddmma_make_dual_feasible": (_I, [_V, _V, _I]),
    "bddmma_gradient_step": (_I, [_V, _V, _D, _I]),
    "bddmma_lbfgs_create": (_I, [C.POINTER(_V), _V, C.POINTER(LbfgsParams)]),
    "bddmma_lbfgs_destroy": (None, [_V]),
    "bddmma_lbfgs_iteration": (_I, [_V]),
    "bddmma_lbfgs_update_costs": (_I, [_V, _V, _U64, _V, _U64, _I, _I]),
    "bddmma_lbfgs_flush": (_I, [_V]),
    "bddmma_lbfgs_get_state": (_I, [_V, C.POINTER(LbfgsState)]),
    "bddmma_perturb_primal_costs": (_I, [_V, _V, _D, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), _V, _V, _V]),
    "bddmma_run_solver": (_I, [_V, _V, _U64, _D, _D, _D, _I, C.POINTER(RunResult)]),
    "bddmma_run_solver_host_loop": (_I, [_V, _V, _U64, _D, _D, _D, _I, C.POINTER(RunResult)]),
    "bddmma_incremental_mm_agreement_rounding": (_I, [_V, _V, _D, _D, _U64, _U64, C.c_uint32, _I, _V, C.POINTER(_I)]),
    "bddmma_save": (_I, [_V, C.c_char_p]),
    "bddmma_load": (_I, [C.POINTER(_V), _I, C.c_char_p]),
    "bddmma_synchronize": (_I, [_V]),
    "bddmma_set_profiling": (_I, [_V, _I]),
    "bddmma_get_profile": (_I, [_V, C.POINTER(Profile)]),
    "bddmma_time_iterations": (_I, [_V, _D, _U64, C.POINTER(_D)]),
    "bddmma_time_kernel": (_I, [_V, _I, _U64, C.POINTER(_D)]),
    "bddmma_device_bytes": (_U64, [_V]),
    "bddmma_device_allocated_bytes": (_U64, [_V]),
    "bddmma_layout_create": (_I, [C.POINTER(_V), _V, _V, _U64, C.POINTER(Options)]),
    "bddmma_layout_create_for_chip": (_I, [C.POINTER(_V), _V, _V, _U64, C.POINTER(Options), _I, C.c_uint32, C.c_uint32]),
    "bddmma_layout_destroy": (None, [_V]),
    "bddmma_layout_size": (_U64, [_V, _I]),
    "bddmma_layout_copy": (_I, [_V, _I, _V]),
    "bddmma_layout_res2_records": (_I, [_V, _I, _V, _V, _V]),
    "bddmma_layout_stream_records": (_I, [_V, _I, _V, _V, _V]),
    "bddmma_layout_layer_records": (_I, [_V, _I, _V, _V, _V]),
    "bddmma_layout_seg_exchange": (_I, [_V, _I, _I, _V, _V, _V, _V]),
}

# every symbol include/bdd_ilp.h declares (host-side input stage)
_I64P, _U64P = C.POINTER(C.c_int64), C.POINTER(C.c_uint64)
ILP_SIGNATURES = {
    "bddilp_last_error": (C.c_char_p, []),
    "bddilp_parse_lp": (_I, [C.c_char_p, C.POINTER(_V)]),
    "bddilp_parse_opb": (_I, [C.c_char_p, C.POINTER(_V)]),
    "bddilp_parse": (_I, [C.c_char_p, C.POINTER(_V)]),
    "bddilp_destroy": (None, [_V]),
    "bddilp_nr_variables": (_U64, [_V]),
    "bddilp_nr_constraints": (_U64, [_V]),
    "bddilp_variable_name": (C.c_char_p, [_V, _U64]),
    "bddilp_objective": (_I, [_V, _V, C.POINTER(_D)]),
    "bddilp_constraint_size": (_U64, [_V, _U64]),
    "bddilp_constraint": (_I, [_V, _U64, _V, _V, C.POINTER(_I), C.POINTER(C.c_int64)]),
    "bddilp_constraint_name": (C.c_char_p, [_V, _U64]),
    "bddilp_normalize": (_I, [_V]),
    "bddilp_to_bdds": (_I, [_V, _I, _U64, C.POINTER(_V)]),
    "bddilp_bdds_create": (_I, [C.POINTER(_V)]),
    "bddilp_bdds_add_row": (_I, [_V, _V, _V, _U64, _I, C.c_int64, C.POINTER(_I)]),
    "bddilp_bdds_split": (_I, [_V, _U64, _U64, _I, _U64P, _U64P]),
    "bddilp_bdds_destroy": (None, [_V]),
    "bddilp_bdds_nr_bdds": (_U64, [_V]),
    "bddilp_bdds_nr_instructions": (_U64, [_V]),
    "bddilp_bdds_nr_variables": (_U64, [_V]),
    "bddilp_bdds_instructions": (_V, [_V]),
    "bddilp_bdds_delimiters": (_V, [_V]),
    "bddilp_write_bdd_lp": (_I, [_V, _V, _U64, _V, _U64, C.c_char_p]),
    "bddilp_export_graphviz": (_I, [_V, _V, _U64, _U64, C.c_char_p]),
    "bddilp_random_set_cover": (_I, [_U64, _U64, _U64, _U64, _V, _V]),
}

_lib = None


def build(verbose: bool = False) -> None:
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    out = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4"], capture_output=True, text=True)
    if out.returncode != 0:
        raise RuntimeError("building libbdd_mma_hip.so failed:\n" + out.stdout + out.stderr)
    if verbose:
        print(out.stdout)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `make -C bdd_amd/csrc` (or __graft_entry__.build()); "
                               "bdd_amd has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SIGNATURES.items()) + list(ILP_SIGNATURES.items()):
            f = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


class BddMmaError(RuntimeError):
    pass


def check(rc: int, handle=None):
    if rc != OK:
        msg = lib().bddmma_last_error(handle)
        raise BddMmaError(f"bdd_mma error {rc}: {msg.decode() if msg else ''}")
