"""Deterministic synthetic instances for tests and the benchmark.

Every generator returns ``(BddCollection, costs)`` or an ``ILP``; nothing here needs the
reference.  Cost tables of the small known-answer problems are the numbers of the reference's
own tests (cited per function); the problems are re-generated from those numbers.
"""
from __future__ import annotations

import numpy as np

from .bdd_collection import BddCollection
from .ilp import ILP


def random_set_cover(n_vars: int, n_rows: int, k: int = 10, seed: int = 12345):
    """Random set cover: each row = `k` distinct variables drawn uniformly, sorted; constraint
    sum x >= 1 -> covering QBDD (2k-1 nodes + 2 terminals); cost U(1,10) for covered variables,
    0 for variables in no row (SURVEY.md §8d "Synthetic inputs").  RNG: numpy PCG64(seed)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rows = rng.integers(0, n_vars, size=(n_rows, k), dtype=np.int64)
    rows.sort(axis=1)
    while True:
        dup = (rows[:, 1:] == rows[:, :-1]).any(axis=1)
        nd = int(dup.sum())
        if nd == 0:
            break
        rows[dup] = rng.integers(0, n_vars, size=(nd, k), dtype=np.int64)
        rows[dup] = np.sort(rows[dup], axis=1)
    costs = rng.uniform(1.0, 10.0, size=n_vars)
    covered = np.zeros(n_vars, dtype=bool)
    covered[rows.ravel()] = True
    costs[~covered] = 0.0
    col = BddCollection()
    col.add_covering(rows.astype(np.uint64))
    return col, costs


def random_set_cover_mt(n_vars: int, n_rows: int, k: int = 10, seed: int = 12345):
    """The benchmark instance: random set cover drawn from std::mt19937_64(seed) by the host library
    (bddilp_random_set_cover, bdd_amd/csrc/host/instances.cpp — the draw order is documented there), so that the C++
    command line and Python produce the same rows.  Same shape as random_set_cover."""
    from . import capi
    rows = np.zeros((n_rows, k), np.uint64)
    costs = np.zeros(n_vars, np.float64)
    rc = capi.lib().bddilp_random_set_cover(n_vars, n_rows, k, seed, rows.ctypes.data, costs.ctypes.data)
    if rc != 0:
        raise ValueError("bddilp_random_set_cover: invalid arguments")
    col = BddCollection()
    col.add_covering(rows)
    return col, costs


def random_set_cover_mixed(n_vars: int, n_rows: int, k_min: int = 3, k_max: int = 16, seed: int = 12345):
    """Set cover with row sizes drawn uniformly from [k_min, k_max], rows of all sizes interleaved at random:
    the structure-heterogeneous counterpart of random_set_cover (BDDs of k_max - k_min + 1 different shapes)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    ks = rng.integers(k_min, k_max + 1, size=n_rows)
    col = BddCollection()
    covered = np.zeros(n_vars, dtype=bool)
    for k in range(k_min, k_max + 1):
        cnt = int((ks == k).sum())
        if cnt == 0:
            continue
        rows = np.sort(rng.integers(0, n_vars, size=(cnt, k), dtype=np.int64), axis=1)
        while True:
            dup = (rows[:, 1:] == rows[:, :-1]).any(axis=1)
            if not dup.any():
                break
            rows[dup] = np.sort(rng.integers(0, n_vars, size=(int(dup.sum()), k), dtype=np.int64), axis=1)
        covered[rows.ravel()] = True
        col.add_covering(rows.astype(np.uint64))
    col.permute(rng.permutation(n_rows))
    costs = rng.uniform(1.0, 10.0, size=n_vars)
    costs[~covered] = 0.0
    return col, costs


def set_cover_sizes(n_vars: int, n_rows: int, k: int):
    """(N, N', L', V, B, H) of random_set_cover in the notation of SURVEY.md §8."""
    return dict(N=n_rows * (2 * k + 1), N_nt=n_rows * (2 * k - 1), L_nt=n_rows * k, V=n_vars, B=n_rows, H=k)


def assignment_ilp(n: int, costs=None) -> ILP:
    """n x n bipartite matching: 2n simplex rows over n^2 variables.  Default costs follow
    test/test_bdd_bipartite_matching_problem.cpp:8-22 / test_bdd_cuda_base.cpp:36-47
    (-2 on the diagonal, -1 elsewhere => optimum -2n)."""
    ilp = ILP()
    if costs is None:
        costs = -np.ones((n, n)) - np.eye(n)
    for i in range(n):
        for j in range(n):
            ilp.objective[ilp.var(f"x_{i + 1}{j + 1}" if n < 10 else f"x_{i + 1}_{j + 1}")] = float(costs[i][j])
    name = (lambda i, j: f"x_{i + 1}{j + 1}") if n < 10 else (lambda i, j: f"x_{i + 1}_{j + 1}")
    for i in range(n):
        ilp.add_constraint([(1, name(i, j)) for j in range(n)], "=", 1)
    for j in range(n):
        ilp.add_constraint([(1, name(i, j)) for i in range(n)], "=", 1)
    return ilp


def mrf_ilp(unaries, edges) -> ILP:
    """Binary pairwise MRF in the local-polytope form of test/test_problems.h:4-195.
    unaries: list of (c0, c1); edges: list of (i, j, (c00, c01, c10, c11)).
    Rows: mu_i_0 + mu_i_1 = 1; sum of the 4 pairwise = 1; marginalisation
    mu_i_a - mu_ij_a0 - mu_ij_a1 = 0 and mu_j_b - mu_ij_0b - mu_ij_1b = 0."""
    ilp = ILP()
    for i, (c0, c1) in enumerate(unaries):
        ilp.objective[ilp.var(f"mu_{i}_0")] = float(c0)
        ilp.objective[ilp.var(f"mu_{i}_1")] = float(c1)
    for i, j, c in edges:
        for ab, cc in zip(("00", "01", "10", "11"), c):
            ilp.objective[ilp.var(f"mu_{i}{j}_{ab}")] = float(cc)
    for i in range(len(unaries)):
        ilp.add_constraint([(1, f"mu_{i}_0"), (1, f"mu_{i}_1")], "=", 1)
    for i, j, _ in edges:
        ilp.add_constraint([(1, f"mu_{i}{j}_00"), (1, f"mu_{i}{j}_10"), (1, f"mu_{i}{j}_01"), (1, f"mu_{i}{j}_11")], "=", 1)
    for i, j, _ in edges:
        for a in (0, 1):
            ilp.add_constraint([(1, f"mu_{i}_{a}"), (-1, f"mu_{i}{j}_{a}0"), (-1, f"mu_{i}{j}_{a}1")], "=", 0)
        for b in (0, 1):
            ilp.add_constraint([(1, f"mu_{j}_{b}"), (-1, f"mu_{i}{j}_0{b}"), (-1, f"mu_{i}{j}_1{b}")], "=", 0)
    return ilp


# Cost tables of the reference's known-answer MRF problems (test/test_problems.h).
# short_mrf_chain (:4-17): LB 1 (test_bdd_cuda_parallel_mma.cu:226 uses the shuffled twin, same costs up to naming)
SHORT_CHAIN = dict(unaries=[(2, 1), (-1, 0)], edges=[(0, 1, (1, 1, 2, 0))])
# long_mrf_chain (:35-100): LB -9
LONG_CHAIN = dict(
    unaries=[(2, -1), (3, -1), (3, 2), (-1, -2), (-2, -1), (1, -1), (1, 1), (-3, 2), (0, 2)],
    edges=[(0, 1, (1, -2, 2, -1)), (1, 2, (0, -1, 1, 0)), (2, 3, (-1, 2, 1, -2)), (3, 4, (2, 0, 2, 2)),
           (4, 5, (1, -2, -3, -1)), (5, 6, (-2, 0, 1, 3)), (6, 7, (-1, -2, -1, -1)), (7, 8, (2, 0, 2, 3))])
# mrf_grid_graph_3x3 (:102-188): LB -8
GRID_3X3 = dict(
    unaries=[(2, -1), (3, -1), (3, 2), (-1, -2), (-2, -1), (3, -1), (1, 1), (-3, 2), (0, 2)],
    edges=[(0, 1, (1, -2, 2, -1)), (1, 2, (0, 1, 1, 0)), (0, 3, (-1, 2, 0, -2)), (1, 4, (2, 0, 2, 2)),
           (2, 5, (1, -2, -3, -1)), (3, 4, (0, 1, 1, 1)), (4, 5, (-1, -2, 4, -2)), (3, 6, (-2, 0, 1, 3)),
           (4, 7, (3, -2, -2, -1)), (5, 8, (0, 1, 1, 1)), (6, 7, (-1, 2, -1, -1)), (7, 8, (2, 0, 2, 2))])


def brute_force_optimum(ilp: ILP) -> float:
    """Exhaustive optimum of a tiny ILP (<= 22 variables); used as an independent check."""
    n = ilp.nr_variables()
    assert n <= 22
    best = np.inf
    obj = np.asarray(ilp.objective)
    for m in range(1 << n):
        x = [(m >> i) & 1 for i in range(n)]
        if ilp.feasible(x):
            best = min(best, float(np.dot(obj, x)))
    return best + ilp.constant
