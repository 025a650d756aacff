"""CPU oracle of the L-BFGS outer loop around parallel MMA.

TEST INFRASTRUCTURE ONLY (see oracle/mma_oracle.c): imported by tests/ only.

Restates the *intended* maths of LPMP::lbfgs<SOLVER, VECTOR, REAL, INT_VECTOR, CUDA_SOLVER>
(reference: include/bdd_solver/lbfgs.h:35-111, src/bdd_solver/lbfgs_impl.h) on top of the MMA oracle
(oracle/mma_oracle.c), function by function:

    store_iterate               lbfgs_impl.h:45-135
    iteration                   :137-157
    search_step_size_and_apply  :159-224
    compute_update_direction    :226-316
    flush_lbfgs_states          :318-326
    lbfgs_update_possible       :334-340
    update_costs                :343-364
    mma_iteration / lbfgs_iteration / choose_solver   :366-417

Parity pinning: UNPINNED against the reference, by necessity — the reference's GPU branches are compiled out
(`#ifdef CUDACC`, never defined, :59,74,89,126,236,254,267,283,303) and on both CPU and GPU the alpha pushed to
`alpha_history` is an uninitialised outer variable shadowed by the one that is computed (:251-263), so the
reference's own trajectory is undefined behaviour.  What is restated is what those lines
intend: the standard two-loop recursion with the initial Hessian diagonal folded into the first beta.  The oracle
is what pins the HIP implementation (bdd_amd/csrc/lbfgs.hip): same solver-selection sequence, same number of
step-size trials, same step sizes and the same lower-bound trajectory.

Solver primitives follow the GPU solver where CPU and GPU differ: x = net_solver_costs() is
hi - lo + the layer's own deferred min-marginal difference (bdd_cuda_parallel_mma.cu:432-463).

Scalars the reference types as REAL (rho_inv, alpha, beta, lb_pre, the relative change, the best step) are kept in
double here and in lbfgs.hip, and dot products are accumulated in double: a sequential float accumulation over
millions of entries (std::inner_product with a float init, :103,259,275,299) would make every decision of the
state machine depend on the summation order.
"""
from __future__ import annotations

from collections import deque

import numpy as np


class LbfgsOracle:
    def __init__(self, mma, history_size=5, init_step_size=1e-6, req_rel_lb_increase=1e-6,
                 step_size_decrease_factor=0.8, step_size_increase_factor=1.1):
        # lbfgs_impl.h:27-31
        assert init_step_size > 0 and 0 < step_size_decrease_factor < 1 and step_size_increase_factor > 1
        assert req_rel_lb_increase > 0 and history_size > 1
        self.s = mma
        self.m = history_size
        self.step_size = float(init_step_size)
        self.req = float(req_rel_lb_increase)
        self.dec = float(step_size_decrease_factor)
        self.inc = float(step_size_increase_factor)
        self.history = deque()          # entries (s: REAL[L], y: int8[L], rho_inv)
        self.prev_x = None
        self.prev_g = None
        self.prev_states_stored = False
        self.num_unsuccessful = 0
        self.lb_history = []
        # diagnostics compared with bddmma_lbfgs_get_state
        self.last_kind = 0              # 0 mma iteration, 1 lbfgs iteration
        self.last_trials = 0            # gradient-step trials of the last step-size search
        self.last_applied_step = 0.0    # step left applied by the last search
        self.mma_iterations = 0
        self.lbfgs_iterations = 0

    @staticmethod
    def _dot(a, b):
        return float(np.dot(a.astype(np.float64), b.astype(np.float64)))

    # lbfgs_impl.h:45-135
    def store_iterate(self, cur_g):
        cur_x = self.s.net_solver_costs()
        if not self.prev_states_stored:
            self.prev_x, self.prev_g = cur_x, cur_g.copy()
            self.prev_states_stored = True
            return
        cur_s = cur_x - self.prev_x                                   # x_k - x_{k-1}, in REAL (:81)
        cur_y = (self.prev_g.astype(np.int8) - cur_g.astype(np.int8))  # grad_{k-1} - grad_k in {-1,0,1} (:100)
        rho_inv = self._dot(cur_s, cur_y)
        if rho_inv > 1e-8:                                            # :112
            self.history.append((cur_s, cur_y, rho_inv))
            if len(self.history) > self.m:
                self.history.popleft()
        else:
            self.prev_states_stored = False
        self.prev_x, self.prev_g = cur_x, cur_g.copy()

    # :334-340
    def update_possible(self):
        return len(self.history) >= self.m and self.num_unsuccessful <= 5

    # :226-316 in the form lbfgs.hip computes it (round 3): the direction is always g + sum a_p y_p + sum b_p s_p, so the two loops
    # run on the coefficients with the dot products <s_i, q>, <y_i, q> expanded over the Gram matrices SS, SY, YY and the products
    # with g ("vector-free L-BFGS", Chen et al. 2014) — 2 passes over the history instead of 2 m dependent ones; the values are
    # those of compute_update_direction_two_loop up to rounding (float: q is not rounded to REAL after every axpy but once at the end).
    def compute_update_direction(self, cur_g):
        dtype = self.s.dtype
        f8 = np.float64
        S = [h[0].astype(f8) for h in self.history]
        Y = [h[1].astype(f8) for h in self.history]
        g = cur_g.astype(f8)
        m = len(S)
        SS = np.array([[np.dot(S[i], S[j]) for j in range(m)] for i in range(m)])
        SY = np.array([[np.dot(S[i], Y[j]) for j in range(m)] for i in range(m)])   # SY[i][j] = <s_i, y_j>; the diagonal is rho_inv
        YY = np.array([[np.dot(Y[i], Y[j]) for j in range(m)] for i in range(m)])
        Sg = np.array([np.dot(S[i], g) for i in range(m)])
        Yg = np.array([np.dot(Y[i], g) for i in range(m)])
        rho_inv = [h[2] for h in self.history]                                       # as stored (== SY[i][i] up to summation order)
        cy, cs = np.zeros(m), np.zeros(m)
        alphas = np.zeros(m)
        for i in range(m - 1, -1, -1):
            dot = Sg[i]
            for r in range(m):
                dot += cy[r] * SY[i][r]
            for r in range(m):
                dot += cs[r] * SS[i][r]
            alphas[i] = dot / rho_inv[i]
            cy[i] -= alphas[i]
        h_diag = rho_inv[m - 1] / (1e-8 + YY[m - 1][m - 1])                           # :291
        for i in range(m):
            rho = 1.0 / rho_inv[i]
            if i == 0:
                rho *= h_diag
            dot = Yg[i]
            for r in range(m):
                dot += cy[r] * YY[i][r]
            for r in range(m):
                dot += cs[r] * SY[r][i]
            cs[i] += alphas[i] - rho * dot
        d = g.copy()
        for i in range(m):
            d = d + cy[i] * Y[i]
        for i in range(m):
            d = d + cs[i] * S[i]
        return np.ascontiguousarray(d.astype(dtype))

    # the literal two-loop recursion (what rounds 1-2 implemented; kept as the cross-check of the form above, tests/test_lbfgs_oracle_forms.py)
    def compute_update_direction_two_loop(self, cur_g):
        dtype = self.s.dtype
        d = cur_g.astype(dtype)
        alphas = []
        for (s_i, y_i, rho_inv) in reversed(self.history):
            alpha = self._dot(s_i, d) / rho_inv
            alphas.append(alpha)
            d = d - dtype(alpha) * y_i.astype(dtype)                  # direction[j] -= alpha * y[j]
        alphas.reverse()
        last_y = self.history[-1][1]
        last_y_norm = self._dot(last_y, last_y)
        h_diag = self.history[-1][2] / (1e-8 + last_y_norm)           # :291
        for i, (s_i, y_i, rho_inv) in enumerate(self.history):
            rho = 1.0 / rho_inv
            if i == 0:
                rho *= h_diag
            beta = rho * self._dot(y_i, d)
            d = d + dtype(alphas[i] - beta) * s_i                     # :311
        return np.ascontiguousarray(d, dtype=dtype)

    # :159-224
    def search_step_size_and_apply(self, update):
        lb_pre = self.s.lower_bound()
        m = self.m

        def rel_change():
            cur_inc = self.s.lower_bound() - lb_pre
            past_inc = self.lb_history[-(m - 1)] - self.lb_history[-m]   # *(rbegin+m-2) - *(rbegin+m-1)
            return cur_inc / (1e-9 + past_inc)

        prev = [0.0]

        def apply(new_step):
            net = new_step - prev[0]
            if net != 0.0:
                self.s.gradient_step(update, net)
                self.last_trials += 1
            prev[0] = new_step

        self.last_trials = 0
        num_updates = 0
        best_step, best_impr = 0.0, 0.0
        while True:
            apply(self.step_size)
            cur = rel_change()
            if best_impr < cur:
                best_impr, best_step = cur, self.step_size
            if cur <= 0.0:
                self.step_size *= self.dec
            elif cur < self.req:
                self.step_size *= self.inc
            if num_updates > 5:
                if best_impr > self.req / 10.0:
                    apply(best_step)
                else:
                    apply(0.0)
                    self.num_unsuccessful += 1
                self.last_applied_step = prev[0]
                return
            num_updates += 1
            if not cur < self.req:
                break
        if num_updates == 1 and self.num_unsuccessful == 0:
            self.step_size *= self.inc
        self.num_unsuccessful = 0
        self.last_applied_step = prev[0]

    # :318-326 and :343-364
    def flush(self):
        self.num_unsuccessful = 0
        self.history.clear()
        self.prev_states_stored = False

    def update_costs(self, lo, hi):
        self.flush()
        self.s.update_costs(lo, hi)

    # :137-157, :366-417
    def iteration(self):
        if not self.lb_history:
            self.lb_history.append(self.s.lower_bound())
        cur_g = self.s.bdds_solution_vec()
        self.store_iterate(cur_g)
        if self.update_possible():
            d = self.compute_update_direction(cur_g)
            self.s.make_dual_feasible(d)
            self.search_step_size_and_apply(d)
            self.s.iteration()
            self.last_kind = 1
            self.lbfgs_iterations += 1
        else:
            self.s.iteration()
            self.last_kind = 0
            self.last_trials = 0
            self.last_applied_step = 0.0
            self.mma_iterations += 1
        self.lb_history.append(self.s.lower_bound())

    def lower_bound(self):
        return self.s.lower_bound()
