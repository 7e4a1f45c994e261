"""CPU: the exact Nash solver behind oak_amd.search.solve_matrix (reference: LRSNash::solve_fast through
pyoak.solve_matrix, cpp/src/pyoak.cc:394-426; no reference test pins it -> checked by LP optimality:
with exact rationals the best-response gap must be exactly zero)."""
from fractions import Fraction

import numpy as np
import pytest

from oak_amd.search import solve_matrix, solve_matrix_exact


def _check_equilibrium(A):
    p1, p2, v = solve_matrix_exact(A)
    m, n = len(A), len(A[0])
    assert sum(p1) == 1 and sum(p2) == 1 and all(x >= 0 for x in p1) and all(x >= 0 for x in p2)
    row_payoffs = [sum(Fraction(A[i][j]) * p2[j] for j in range(n)) for i in range(m)]
    col_payoffs = [sum(Fraction(A[i][j]) * p1[i] for i in range(m)) for j in range(n)]
    assert max(row_payoffs) == v and min(col_payoffs) == v        # zero best-response gap, exactly
    return p1, p2, v


def test_rock_paper_scissors_and_saddle_points():
    p1, p2, v = _check_equilibrium([[128, 0, 256], [256, 128, 0], [0, 256, 128]])
    assert p1 == [Fraction(1, 3)] * 3 and p2 == [Fraction(1, 3)] * 3 and v == 128
    p1, p2, v = _check_equilibrium([[3, 5], [1, 0]])               # saddle at (0, 0)
    assert p1 == [1, 0] and p2 == [1, 0] and v == 3
    p1, p2, v = _check_equilibrium([[7]])
    assert v == 7
    p1, p2, v = _check_equilibrium([[0, 256], [256, 0]])           # matching pennies
    assert v == 128 and p1 == [Fraction(1, 2)] * 2


def test_random_matrices_have_zero_best_response_gap():
    rng = np.random.default_rng(0)
    for _ in range(300):
        m, n = rng.integers(1, 10, 2)
        A = rng.integers(0, 257, (m, n)).tolist()
        _check_equilibrium(A)
    for _ in range(50):                                             # degenerate: many ties
        m, n = rng.integers(1, 10, 2)
        A = rng.integers(0, 3, (m, n)).tolist()
        _check_equilibrium(A)


def test_solve_matrix_api_and_errors():
    p1, p2, value = solve_matrix(np.array([[256, 0], [0, 256]]), 256)
    assert abs(value - 0.5) < 1e-12 and np.allclose(p1, [.5, .5]) and np.allclose(p2, [.5, .5])
    with pytest.raises(RuntimeError):
        solve_matrix(np.zeros((10, 2), dtype=int), 256)
