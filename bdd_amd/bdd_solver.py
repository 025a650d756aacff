"""`bdd_solver` — the reference's orchestrator and JSON config surface, re-hosted thinly over the HIP backend.

Mirrors `LPMP::bdd_solver` (reference: include/bdd_solver/bdd_solver.h:45-103, src/bdd_solver/bdd_solver.cpp:36-527)
and the pybind class `bdd_solver_py.bdd_solver` (src/bdd_solver/bdd_solver_py.cpp:9-20: ctor from a config
string or dict, `solve()`, `lower_bound()`, `min_marginals()`).  Pipeline, as `bdd_solver::solve` (:477-495):
read_ILP -> process_ILP -> transform_to_BDDs -> construct_solver -> solve_dual -> perturbation_rounding.

Only the relaxation solvers of the hot path exist here (`cuda parallel mma` and the GPU L-BFGS); the CPU
solvers, variable re-orderings and the exporters other than `.lp` belong to parts of the
reference that SURVEY.md §2 marks out of scope — asking for them raises the same kind of
`RuntimeError` the reference raises for an unknown option.
"""
from __future__ import annotations

import json
import os
import time

import numpy as np

from . import capi, native
from .ilp import ILP, parse_lp, parse_lp_or_opb, parse_opb
from .solver import bdd_hip_lbfgs, bdd_hip_parallel_mma, run_solver

GPU_MMA = {"cuda parallel mma", "hip parallel mma"}
GPU_LBFGS = {"lbfgs cuda mma", "cuda lbfgs parallel mma", "lbfgs hip mma", "hip lbfgs parallel mma"}  # bdd_solver.cpp:222,251; README.md:30,59
CPU_ONLY = {"sequential mma", "parallel mma", "lbfgs parallel mma", "subgradient"}


def _log(msg, quiet):
    if not quiet:
        print(msg, flush=True)


class bdd_solver:
    def __init__(self, config, quiet: bool = False):
        self.config = self.read_config(config)
        self.quiet = quiet
        self.ilp: ILP = None
        self.bdd_col = None
        self.solver = None       # bdd_hip_parallel_mma
        self.lbfgs = None        # bdd_hip_lbfgs or None
        self.solution = None     # list of 0/1 after a successful rounding
        self.result = None

    # ------------------------------------------------------------------ config (bdd_solver.cpp:468-475)
    @staticmethod
    def read_config(c):
        if isinstance(c, dict):
            return dict(c)
        if isinstance(c, str) and os.path.exists(c):
            with open(c) as f:
                return json.load(f)
        return json.loads(c)

    # ------------------------------------------------------------------ read_ILP (:44-66)
    def read_ILP(self) -> ILP:
        if "input" not in self.config:
            raise RuntimeError("no input specified")
        inp = self.config["input"]
        if os.path.exists(inp):
            _log(f"[bdd_solver] Read input file {inp}", self.quiet)
            with open(inp) as f:
                return parse_opb(f.read()) if os.path.splitext(inp)[1] == ".opb" else parse_lp_or_opb(f.read())
        _log("[bdd_solver] Read input string", self.quiet)
        return parse_lp_or_opb(inp)      # the reference tries the .lp grammar, then OPB (:59-63)

    # ------------------------------------------------------------------ process_ILP (:71-103)
    def process_ILP(self, ilp: ILP):
        order = self.config.get("variable order", "input")
        if order in ("bfs", "cuthill", "minimum degree"):
            raise RuntimeError(f"Variable order {order} is not available in this backend (ILP re-orderings are outside the hot path)")
        if order != "input":
            raise RuntimeError(f"Variable order {order} unknown")
        if self.config.get("normalize constraints", False):
            _log("[bdd_solver] Normalize constraints", self.quiet)
            for c in ilp.constraints:  # ILP_input::constraint::normalize: monomials sorted by variable
                idx = np.argsort(c.variables, kind="stable")
                c.variables = [c.variables[i] for i in idx]
                c.coefficients = [c.coefficients[i] for i in idx]

    # ------------------------------------------------------------------ transform_to_BDDs (:112-123)
    def transform_to_BDDs(self, ilp: ILP):
        _log("[bdd solver] Compute BDDs", self.quiet)
        # conversion and splitting run in the C++ input stage (bdd_amd/csrc/host/, include/bdd_ilp.h);
        # ilp.to_bdd_collection / ilp.split_long_bdds are the same algorithms in Python (tests/test_native_host.py)
        rows = [(c.coefficients, c.variables, c.ineq, c.rhs) for c in ilp.constraints]
        split_length, implication = None, False
        if "split bdds" in self.config:
            sb = self.config["split bdds"] or {}
            # the reference tests contains("implication bdd") and then reads key "implication" (:119); accept both
            implication = bool(sb.get("implication bdd", sb.get("implication", False)))
            split_length = int(sb.get("split length", 0))       # 0: the occupancy rule of compute_split_length
        col = native.rows_to_bdd_collection(rows, split_length=split_length, nr_variables=ilp.nr_variables(),
                                            with_implication_bdd=implication)
        _log(f"[bdd preprocessor] final #BDDs = {col.nr_bdds()}", self.quiet)
        return col

    # ------------------------------------------------------------------ construct_solver (:130-267)
    def construct_solver(self, bdd_col, costs):
        precision = self.config.get("precision", "double")
        if precision not in ("double", "single", "float"):
            raise RuntimeError("precision must be double|single|float")
        name = self.config.get("relaxation solver", "cuda parallel mma")
        if name in CPU_ONLY:
            raise RuntimeError(f"relaxation solver {name} is a CPU solver of the reference; this backend provides "
                               f"{sorted(GPU_MMA | GPU_LBFGS)}")
        if name not in GPU_MMA | GPU_LBFGS:
            raise RuntimeError(f"relaxation solver {name} unknown")
        # NB: the reference constructs the <float> GPU solver for "double" and vice versa (:167-174); here
        # "precision" means what it says.
        # The solver's variables are those of the BDDs.  A variable that occurs in the objective only is free: its better value
        # contributes min(0, c) to the bound and is fixed in the primal (the reference only asserts on such costs,
        # bdd_parallel_mma_base.cpp:685-695).
        # Such a variable may have ANY index: the LP reader numbers variables by first appearance and reads the objective first,
        # so an objective-only variable usually lies below the last constrained one — it is recognised by its BDD
        # count, not by its position.
        nv = bdd_col.nr_variables()
        costs = np.asarray(costs, float)
        c = np.zeros(nv)
        c[: min(nv, len(costs))] = costs[:nv]
        s = bdd_hip_parallel_mma(bdd_col, c, precision="double" if precision == "double" else "float",
                                 device=int(self.config.get("device", 0)))
        in_bdd = np.zeros(len(costs), bool)
        in_bdd[: min(nv, len(costs))] = s.get_num_bdds_per_var()[: len(costs)] > 0
        self.free_vars = np.flatnonzero(~in_bdd)                      # objective-only variables, any index
        self.free_ones = (costs[self.free_vars] < 0).astype(np.int8)  # their better value
        self.free_constant = float(costs[self.free_vars][costs[self.free_vars] < 0].sum())
        lb = None
        if name in GPU_LBFGS:
            p = self.config.get("lbfgs", {})  # :179-199
            lb = bdd_hip_lbfgs(s, history_size=p.get("history size", 5), init_step_size=p.get("initial step size", 1e-6),
                               req_rel_lb_increase=p.get("required relative lb increase", 1e-6),
                               step_size_decrease_factor=p.get("step size decrease factor", 0.8),
                               step_size_increase_factor=p.get("step size increase factor", 1.1))
        return s, lb

    # ------------------------------------------------------------------ solve_dual (:277-309)
    def solve_dual(self):
        tc = self.config.get("termination criteria", {})
        max_iter = int(tc.get("maximum iterations", 1000))
        min_improvement = float(tc.get("minimum improvement", 1e-6))
        improvement_slope = float(tc.get("improvement slope", 1e-9))
        time_limit = float(tc.get("time limit", 3600))
        self.result = run_solver(self.solver, max_iter, min_improvement, improvement_slope, time_limit,
                                 verbose=not self.quiet, lbfgs=self.lbfgs)
        _log("[bdd solver] Terminated dual optimization", self.quiet)
        return self.result

    # ------------------------------------------------------------------ perturbation_rounding (:318-380)
    def perturbation_rounding(self):
        if "perturbation rounding" not in self.config:
            return []
        pr = self.config["perturbation rounding"]
        import ctypes as C
        L = capi.lib()
        V = self.solver.nr_variables()
        sol = np.zeros(V, np.int8)
        found = C.c_int(0)
        rc = L.bddmma_incremental_mm_agreement_rounding(
            self.solver._h, self.lbfgs._h if self.lbfgs else None, float(pr.get("initial perturbation", 0.1)),
            float(pr.get("perturbation growth rate", 1.1)), int(pr.get("inner iterations", 100)),
            int(pr.get("outer iterations", 100)), int(pr.get("seed", 0)), 0 if self.quiet else 1,
            sol.ctypes.data_as(C.c_void_p), C.byref(found))
        capi.check(rc, self.solver._h)
        if not found.value:
            _log("[incremental primal rounding] No solution found", self.quiet)
            return []
        full = np.zeros(max(V, self.ilp.nr_variables()), np.int8)
        full[:V] = sol
        keep = self.free_vars < len(full)
        full[self.free_vars[keep]] = self.free_ones[keep]   # variables of the objective only: their better value
        self.solution = full[: self.ilp.nr_variables()].astype(int).tolist()
        obj = self.ilp.evaluate(self.solution) if self.ilp.feasible(self.solution) else float("inf")
        _log(f"[incremental primal rounding] solution objective = {obj}", self.quiet)
        return self.solution

    # ------------------------------------------------------------------ solve (:477-495)
    def solve(self):
        if self.ilp is None:
            t0 = time.time()
            self.ilp = self.read_ILP()
            self.process_ILP(self.ilp)
            self.export_lp()
            self.bdd_col = self.transform_to_BDDs(self.ilp)
            self.print_statistics()
            if "export bdd lp" in self.config:          # bdd_solver.cpp:400-410
                with open(self.config["export bdd lp"], "w") as f:
                    f.write(self.bdd_col.write_bdd_lp(self.ilp.objective))
            if "export bdd graph" in self.config:       # :432-462: <name>_<bdd>.dot per BDD (the reference also shells out to `dot -Tpng`)
                stem = os.path.splitext(self.config["export bdd graph"])[0]
                for b in range(self.bdd_col.nr_bdds()):
                    with open(f"{stem}_{b}.dot", "w") as f:
                        f.write(self.bdd_col.export_graphviz(b))
            self.solver, self.lbfgs = self.construct_solver(self.bdd_col, self.ilp.objective)
            _log(f"[bdd solver] set-up time = {time.time() - t0:.3f} s", self.quiet)
        self.solve_dual()
        self.perturbation_rounding()
        return self

    def lower_bound(self) -> float:
        return self.solver.lower_bound() + self.ilp.constant + getattr(self, "free_constant", 0.0)

    def min_marginals(self):
        """[var][bdd] -> (mm0, mm1).  (The reference throws for GPU solvers, :497-514; the backend has them.)"""
        mm = self.solver.min_marginals()
        return mm[: self.ilp.nr_variables()]

    def min_marginals_with_variable_names(self):
        mm = self.min_marginals()
        return self.ilp.var_names, [m[:, 0].tolist() for m in mm], [m[:, 1].tolist() for m in mm]

    # ------------------------------------------------------------------ small helpers (:382-416)
    def print_statistics(self):
        if "print statistics" not in self.config:
            return
        per_var = np.zeros(self.ilp.nr_variables(), int)
        for c in self.ilp.constraints:
            per_var[list(set(c.variables))] += 1
        print(f"[print_statistics] #variables = {self.ilp.nr_variables()}")
        print(f"[print_statistics] #constraints = {len(self.ilp.constraints)}")
        print(f"[print_statistics] #BDDs = {self.bdd_col.nr_bdds()}")
        print(f"[print_statistics] minimum num. constraints per var = {per_var.min()}")
        print(f"[print_statistics] maximum num. constraints per var = {per_var.max()}")
        print(f"[print_statistics] mean num. constraints per var = {per_var.mean()}")

    def export_lp(self):
        if "export lp" not in self.config:
            return
        path = self.config["export lp"]
        if os.path.splitext(path)[1] != ".lp":
            raise RuntimeError(f"Cannot recognize file extension {os.path.splitext(path)[1]} for exporting problem file")
        with open(path, "w") as f:
            f.write(self.ilp.write_lp())
