"""CPU: the exact Nash solver behind oakgpu_solve_matrix / oak_amd.search.solve_matrix (reference: LRSNash::solve_fast
through pyoak.solve_matrix, cpp/src/pyoak.cc:394-426; no reference test pins it).  Checked two ways: an independent
exact-rational simplex written here (fractions.Fraction) must have a best-response gap of exactly zero, and the C++
solver (integer pivoting on 512-bit integers, same pivoting rule) must return the same vertex and value."""
from fractions import Fraction

import numpy as np
import pytest

from oak_amd.search import solve_matrix


def _simplex_max(A, b, c):
    """maximise c.x s.t. A x <= b, x >= 0 with b > 0; exact Fractions, Bland's rule; returns (x, dual y, value)."""
    m, n = len(A), len(A[0])
    T = [[Fraction(v) for v in A[i]] + [Fraction(int(i == k)) for k in range(m)] + [Fraction(b[i])] for i in range(m)]
    z = [-Fraction(v) for v in c] + [Fraction(0)] * m + [Fraction(0)]
    basis = [n + i for i in range(m)]
    while True:
        col = next((j for j in range(n + m) if z[j] < 0), None)
        if col is None:
            break
        best, row = None, None
        for i in range(m):
            if T[i][col] > 0:
                ratio = T[i][-1] / T[i][col]
                if best is None or ratio < best or (ratio == best and basis[i] < basis[row]):
                    best, row = ratio, i
        assert row is not None
        piv = T[row][col]
        T[row] = [v / piv for v in T[row]]
        for i in range(m):
            if i != row and T[i][col] != 0:
                f = T[i][col]
                T[i] = [a - f * p for a, p in zip(T[i], T[row])]
        f = z[col]
        z = [a - f * p for a, p in zip(z, T[row])]
        basis[row] = col
    x = [Fraction(0)] * n
    for i, bi in enumerate(basis):
        if bi < n:
            x[bi] = T[i][-1]
    return x, [z[n + i] for i in range(m)], z[-1]


def solve_matrix_exact(payoffs):
    """Checker: exact equilibrium (Fractions) of the zero-sum game where the row player maximises payoffs[i][j]."""
    A = [[Fraction(v) for v in row] for row in payoffs]
    m, n = len(A), len(A[0])
    shift = 1 - min(min(r) for r in A)
    B = [[v + shift for v in row] for row in A]
    z, y, tot = _simplex_max(B, [1] * m, [1] * n)
    v = 1 / tot
    return [yi * v for yi in y], [zi * v for zi in z], v - shift


def _check_equilibrium(A):
    p1, p2, v = solve_matrix_exact(A)
    m, n = len(A), len(A[0])
    assert sum(p1) == 1 and sum(p2) == 1 and all(x >= 0 for x in p1) and all(x >= 0 for x in p2)
    row_payoffs = [sum(Fraction(A[i][j]) * p2[j] for j in range(n)) for i in range(m)]
    col_payoffs = [sum(Fraction(A[i][j]) * p1[i] for i in range(m)) for j in range(n)]
    assert max(row_payoffs) == v and min(col_payoffs) == v        # zero best-response gap, exactly
    # the product solver (C++, exact integer pivoting, same rule): same vertex, same value, rounded to double only at the end
    c1, c2, cv = solve_matrix(np.array(A), 1)
    assert np.abs(c1 - np.array([float(x) for x in p1])).max() <= 1e-15
    assert np.abs(c2 - np.array([float(x) for x in p2])).max() <= 1e-15
    assert abs(cv - float(v)) <= 1e-12 * max(1.0, abs(float(v)))
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
    for _ in range(40):                                             # the largest payoffs the ABI accepts, negatives included
        m, n = rng.integers(1, 10, 2)
        A = rng.integers(-(1 << 20), (1 << 20) + 1, (m, n)).tolist()
        _check_equilibrium(A)


def test_solve_matrix_api_and_errors():
    p1, p2, value = solve_matrix(np.array([[256, 0], [0, 256]]), 256)
    assert abs(value - 0.5) < 1e-12 and np.allclose(p1, [.5, .5]) and np.allclose(p2, [.5, .5])
    with pytest.raises(RuntimeError):
        solve_matrix(np.zeros((10, 2), dtype=int), 256)
    with pytest.raises(RuntimeError):
        solve_matrix(np.array([[1 << 21]]), 256)                    # beyond the exact solver's proven range


def test_bandit_arithmetic_matches_reference_traces():
    """SURVEY 8(f) rank 1 pin: select / update traces of all five bandits, dumped from the reference's OWN headers
    (search/bandit/{ucb,pucb,ucb1,exp3,pexp3}.h via oracle/ref_bandit_dump.cc -> tests/golden/bandit_traces.json), replayed
    through the product's bandit (oak_amd/csrc/bandit.hpp, oakgpu_bandit_replay): every selected index, every selection
    probability and the final statistics must agree BIT FOR BIT.  Host-side code: no GPU needed."""
    import ctypes as C
    import json
    import os
    from oak_amd import _lib
    lib = _lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    traces = json.load(open(os.path.join(root, "tests", "golden", "bandit_traces.json")))["traces"]
    kinds = {"ucb": 0, "pucb": 1, "ucb1": 2, "exp3": 3, "pexp3": 4}
    f32 = lambda xs: np.array([float(x) for x in xs], dtype=np.float32)   # "-inf" strings included
    seen = set()
    for t in traces:
        kind, k, steps = kinds[t["kind"]], t["k"], t["steps"]
        values = f32(t["values"])
        logits = f32(t["logits"]) if "logits" in t else None
        uniforms = np.array(t.get("uniforms", []), dtype=np.float64)
        idx = np.zeros(steps, dtype=np.uint8)
        prob = np.zeros(steps, dtype=np.float32)
        stats = np.zeros(18, dtype=np.float32)
        visits = np.zeros(9, dtype=np.uint32)
        P = lambda a: None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)
        rc = lib.oakgpu_bandit_replay(kind, C.c_float(t["c"]), C.c_float(t["alpha"]), k, P(logits), steps, P(uniforms), P(values),
                                      P(idx), P(prob), P(stats), P(visits))
        assert rc == 0
        tag = (t["kind"], k, t["seed"])
        assert list(idx) == t["index"], tag
        if kind >= 3:
            assert prob.tobytes() == f32(t["prob"]).tobytes(), tag
            assert stats[:9].tobytes() == f32(t["gains"]).tobytes(), tag
            assert len(uniforms) == (steps if k > 1 else 0)
        else:
            assert stats[:k].tobytes() == f32(t["scores"]).tobytes(), tag
            assert list(visits[:k]) == t["visits"], tag
            if "priors" in t:
                assert stats[9:9 + k].tobytes() == f32(t["priors"]).tobytes(), tag
        seen.add((t["kind"], k))
    assert seen == {(n, k) for n in kinds for k in (1, 2, 4, 9)}
    # the traces are not degenerate: every 9-armed one visits every arm
    assert all(len(set(t["index"])) == 9 for t in traces if t["k"] == 9)


def test_root_select_run_equals_the_plain_select_visit_loop():
    """Bandit::select_run (csrc/bandit.hpp): the root's batch of consecutive select + visit rounds with the counts kept in SSE
    registers must be the plain select(); visit() loop, index for index -- UCB and PUCB at every arm count, from random
    statistics incl. visit counts beyond 2^24 (where a float counter would stop counting), plus the kinds that take the loop."""
    import ctypes as C
    from oak_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(11)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    for trial in range(120):
        kind = [0, 1, 0, 1, 2, 3][trial % 6]
        k = 1 + trial % 9
        big = trial % 5 == 0
        visits = (rng.integers(1, 1 << 26 if big else 4000, 9)).astype(np.uint32)
        scores = (rng.random(9) * visits * rng.random()).astype(np.float32)
        pri = rng.random(9).astype(np.float32); pri[:k] /= pri[:k].sum()
        if kind == 3:
            scores = (-rng.random(9) * 5).astype(np.float32); scores[np.argmax(scores[:k])] = 0.0; scores[k:] = -np.inf; visits[:] = 0
        count = 3000
        run, loop = np.zeros(count, np.uint8), np.zeros(count, np.uint8)
        vr, vl = np.zeros(9, np.uint32), np.zeros(9, np.uint32)
        rc = lib.oakgpu_bandit_select_run(kind, C.c_float(0.3 + 2 * rng.random()), C.c_float(0.05), k, P(scores), P(pri), P(visits), count,
                                          P(run), P(loop), P(vr), P(vl))
        assert rc == 0
        assert (run == loop).all(), (trial, kind, k, int(np.argmax(run != loop)))
        assert (vr == vl).all() and (run < k).all()
        if kind < 2:
            assert int(vr.sum()) == int(visits.sum()) + (count if k > 1 else 0) or k == 1


@pytest.mark.parametrize("threads", [1, 4, 8, 16])
def test_promoted_heap_keeps_the_shard_invariant_of_the_threaded_walk(threads):
    """Round-3 advice (data race after Heap::update): Tree::keep_subtree kept a node's OLD creator shard while its edge
    re-hashed into another table, so the thread of that table read arena[owner][creator'] while thread creator' could append
    to (and reallocate) it.  Host-only self-test of the sharded tree: grow with the search's own threaded resolve phase,
    promote a child of the root, check that every edge's child sits in an arena of the edge's table, grow again with the
    same threads.  The tree's size does not depend on the number of threads."""
    import ctypes as C
    from oak_amd import _lib
    lib = _lib.load()
    out = (C.c_uint64 * 4)()
    rc = lib.oakgpu_heap_selftest(24, 2048, 7, threads, C.byref(out))
    assert rc == 0, (rc, list(out))
    before, kept, after, bad = list(out)
    assert bad == 0 and 0 < kept < before and after > kept
    ref = (C.c_uint64 * 4)()
    assert lib.oakgpu_heap_selftest(24, 2048, 7, 1, C.byref(ref)) == 0
    assert list(ref) == list(out)
