"""numpy restatement of the deterministic part of the reference's GPU primal rounding step.

TEST INFRASTRUCTURE ONLY (see oracle/mma_oracle.c): imported by tests/ only.

Follows src/bdd_solver/incremental_mm_agreement_rounding_cuda.cu of the reference:

    mm_diff_direction      mm_diff_direction_func              :29-41
    compute_mm_types       compute_mm_types + fill_mm_type_func :43-65, :76-108
    compute_mm_sums        compute_mm_sums                      :110-134
    perturbation           mm_types_transform                   :136-205 (only_perturb_inconsistent = false, as called at :325)
    counts                 perturb_primal_costs                 :262-331

Inputs are what min_marginals_cuda(get_sorted = true) returns (bdd_cuda_base.cu:716-749): per layer the variable and
the two min-marginals, sorted by (variable, bdd).  The `equal` / `inconsistent` types draw a random number from
thrust::default_random_engine discarded by thread id (:177-181); that draw is not restated — for those variables only
the side that receives the perturbation (given the draw's sign for `equal`, by the sums for `inconsistent`) and the
bound |r| * delta <= delta^2 are defined here.

Parity pinning: the classification thresholds and the {delta, 0} / {0, delta} table are checked in
tests/test_rounding_oracle.py against hand-computed cases that follow the reference's functors line by line; the
reference itself needs CUDA / Thrust and cannot run here.
"""
from __future__ import annotations

import numpy as np

ONE, ZERO, EQUAL, INCONSISTENT = 0, 1, 2, 3   # order of the counts: #one, #zero, #equal, #inconsistent


def mm_diff_direction(mm0, mm1):
    """:29-41 — `mm_0 + 1e-6 <= mm_1` etc. with a double literal: the comparison is made in double."""
    a, b = np.asarray(mm0, np.float64), np.asarray(mm1, np.float64)
    return np.where(a + 1e-6 <= b, -1, np.where(b + 1e-6 <= a, 1, 0)).astype(np.int8)


def compute_mm_types(n_vars, var, mm0, mm1):
    """:43-65, :76-108 — min / max of the direction over the layers of a variable, then the type table.
    Variables in no BDD have no entry in the reference's reduce_by_key; the HIP path classifies them `zero`."""
    d = mm_diff_direction(mm0, mm1)
    dmin = np.full(n_vars, 2, np.int8)
    dmax = np.full(n_vars, -2, np.int8)
    np.minimum.at(dmin, var, d)
    np.maximum.at(dmax, var, d)
    t = np.full(n_vars, INCONSISTENT, np.int8)
    t[(dmax == 0) & (dmin == 0)] = EQUAL
    t[dmax < 0] = ZERO
    t[dmin > 0] = ONE        # tested first in fill_mm_type_func
    t[dmin == 2] = ZERO      # variable in no BDD
    return t


def compute_mm_sums(n_vars, var, mm0, mm1, dtype):
    """:110-134 — reduce_by_key sums in REAL, layers of a variable in (variable, bdd) order."""
    s0 = np.zeros(n_vars, dtype)
    s1 = np.zeros(n_vars, dtype)
    # sequential accumulation in REAL, in the sorted order
    for v, a, b in zip(var, np.asarray(mm0, dtype), np.asarray(mm1, dtype)):
        s0[v] = dtype(s0[v] + a)
        s1[v] = dtype(s1[v] + b)
    return s0, s1


def perturbation(types, s0, s1, delta, dtype):
    """:136-205 — cost_delta_0 / cost_delta_1 for the deterministic types; for the random types the side is returned
    where it is defined by the inputs (`inconsistent`: mm_0 < mm_1 -> side 1) and -1 where the draw's sign decides."""
    n = len(types)
    c0 = np.zeros(n, dtype)
    c1 = np.zeros(n, dtype)
    c0[types == ONE] = dtype(delta)
    c1[types == ZERO] = dtype(delta)
    side = np.full(n, -2, np.int8)          # -2: deterministic, -1: sign of the draw, 0 / 1: side fixed by the sums
    side[types == EQUAL] = -1
    inc = types == INCONSISTENT
    side[inc] = np.where(s0[inc] < s1[inc], 1, 0)
    return c0, c1, side


def counts(types):
    return tuple(int((types == k).sum()) for k in (ONE, ZERO, EQUAL, INCONSISTENT))
