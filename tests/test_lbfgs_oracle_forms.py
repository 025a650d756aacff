"""The L-BFGS oracle's direction in its two forms (CPU only): the Gram-matrix ("vector-free") form that lbfgs.hip computes since
round 3 against the literal two-loop recursion of lbfgs_impl.h:226-316 (oracle/lbfgs_oracle.py)."""
import numpy as np
import pytest

from oracle.lbfgs_oracle import LbfgsOracle


class _Mma:
    def __init__(self, dtype):
        self.dtype = dtype


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 2e-5)])
@pytest.mark.parametrize("m", [2, 5, 8])
def test_gram_form_equals_two_loop(dtype, tol, m):
    rng = np.random.Generator(np.random.PCG64(17 + m))
    n = 4000
    o = LbfgsOracle(_Mma(dtype), history_size=m)
    x = rng.normal(0, 1, n)
    for _ in range(m):
        # pairs with positive curvature, y in {-1, 0, 1} as the subgradient differences are
        y = rng.integers(-1, 2, n).astype(np.int8)
        s = (0.3 * y + rng.normal(0, 0.05, n)).astype(dtype)
        rho_inv = float(np.dot(s.astype(np.float64), y.astype(np.float64)))
        assert rho_inv > 1e-8
        o.history.append((s, y, rho_inv))
    g = rng.integers(0, 2, n).astype(np.int8)
    a = o.compute_update_direction(g)
    b = o.compute_update_direction_two_loop(g)
    assert a.dtype == dtype and b.dtype == dtype
    scale = float(np.abs(b).max())
    np.testing.assert_allclose(a, b, rtol=0, atol=tol * scale)
