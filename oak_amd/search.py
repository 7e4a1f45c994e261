"""Nash-at-root search fed by GPU batches (BASELINE config 5) -- host-side consumer of the hot path.

Mirrors two pieces of the reference's search surface:
  * `solve_matrix(payoffs, discretize_factor) -> (p1, p2, value)`  (pyoak: cpp/src/pyoak.cc:394-426; used by
    MCTS::Search::process_output, cpp/include/search/mcts.h:620-659, through the absent lrsnash/GMP library):
    exact Nash equilibrium of a <= 9 x 9 one-sum matrix game with integer payoffs -- a thin call into the C ABI
    (oakgpu_solve_matrix: integer-pivoting simplex, Bland's rule, no floating point in the solve).
  * `root_matrix_search(...)`: for every joint action (i, j) of the root and R replicas, re-seed + resample hidden
    counters (mcts.h:254-259), apply the joint action with ONE batched update on the GPU, evaluate the R x m x n
    children with GPU rollouts (or the GPU network), average into the value matrix and solve it.  Output fields
    follow MCTS::Output (mcts.h:68-90).
  * `tree_search(...)`: the full tree search -- MCTS::Search::run (mcts.h:154-247) with a Node heap and joint
    UCB / PUCB bandits -- as `oakgpu_search` runs it: batches of descents walking a host-side tree, every battle
    state resident on the GPU (oak_amd/csrc/search_host.hip), process_output's Nash solve included.
All battle arithmetic runs through the C ABI (oak_amd.engine); this module only orchestrates and solves.
"""
import ctypes as C

import numpy as np

from . import _lib


def solve_matrix(payoffs, discretize_factor=256):
    """pyoak `solve_matrix` (pyoak.cc:394-426) over oakgpu_solve_matrix (exact integer-pivoting simplex in C++,
    oak_amd/csrc/nash.hpp).  payoffs: m x n integers = value * discretize_factor (the row player's one-sum payoff).
    Returns (p1 float[m], p2 float[n], value float)."""
    P = np.asarray(payoffs)
    if P.ndim != 2 or P.shape[0] < 1 or P.shape[1] < 1 or P.shape[0] > 9 or P.shape[1] > 9:
        raise RuntimeError("solve_matrix: payoff matrix must be between 1x1 and 9x9")
    m, n = P.shape
    M = np.ascontiguousarray(P, dtype=np.int32)
    p1, p2, v = np.zeros(m), np.zeros(n), C.c_double(0)
    _lib.check(_lib.load().oakgpu_solve_matrix(M.ctypes.data_as(C.c_void_p), m, n, int(discretize_factor), p1.ctypes.data_as(C.c_void_p),
                                               p2.ctypes.data_as(C.c_void_p), C.cast(C.byref(v), C.c_void_p)))
    return p1, p2, v.value


def root_matrix_search(ctx, battle, durations, result, replicas=256, seed=0x5EED, evaluator="mc", max_steps=1000):
    """One-ply Nash-at-root search of a position (battle uint8[384], durations uint8[8], result byte).

    evaluator: "mc" (random rollouts on the GPU, MCTS::MonteCarlo) or an oak_amd.engine.Network
    (value_inference on the GPU).  Returns a dict shaped like MCTS::Output (mcts.h:68-90)."""
    battle = np.ascontiguousarray(battle, dtype=np.uint8).reshape(1, 384)
    durations = np.ascontiguousarray(durations, dtype=np.uint8).reshape(1, 8)
    res = np.array([result], dtype=np.uint8)
    c1, n1 = ctx.choices(battle, res, 0)
    c2, n2 = ctx.choices(battle, res, 1)
    m, n = int(n1[0]), int(n2[0])
    lanes = m * n * replicas
    B = np.repeat(battle, lanes, axis=0)
    D = np.repeat(durations, lanes, axis=0)
    R = np.repeat(res, lanes)
    # one fast_prng stream per lane (util/random.h:67-133); seeded counter-style, state never all-zero
    rng = np.random.default_rng(seed)
    prng = rng.integers(0, 256, (lanes, 8), dtype=np.uint8)
    prng[:, 0] |= 1
    # root-iteration prep on the device (mcts.h:254-259): max_steps = 0 -> only re-seed + hidden-variable resampling
    prepped = ctx.rollout(B, D, R, prng, max_steps=0, prep=True, return_state=True)
    B, D, prng = prepped["battles"], prepped["durations"], prepped["prng"]
    ii, jj = np.meshgrid(np.arange(m), np.arange(n), indexing="ij")
    a1 = np.repeat(c1[0, ii.ravel()], replicas)
    a2 = np.repeat(c2[0, jj.ravel()], replicas)
    child_res, _ = ctx.update(B, a1, a2, D, want_actions=False)          # in place on B, D
    t = child_res & 15
    values = np.where(t == 1, 1.0, np.where(t == 2, 0.0, 0.5)).astype(np.float32)
    live = t == 0
    if live.any():
        if evaluator == "mc":
            out = ctx.rollout(B[live], D[live], child_res[live], prng[live], max_steps=max_steps)
            values[live] = out["values"]
        else:
            values[live] = evaluator.value_inference(B[live], D[live])
    cum = values.reshape(m, n, replicas).sum(axis=2).astype(np.float64)
    visits = np.full((m, n), replicas, dtype=np.int64)
    mean = cum / visits
    p1, p2, nash_value = solve_matrix(np.floor(mean * 256).astype(np.int64), 256)   # mcts.h:631-633: value / n * 256 as int
    return {
        "m": m, "n": n, "p1_choices": c1[0, :m].copy(), "p2_choices": c2[0, :n].copy(),
        "visit_matrix": visits, "value_matrix": cum,                       # cumulative, like the reference (mcts.h:80-81)
        "iterations": lanes, "empirical_value": float(values.mean()), "nash_value": nash_value,
        "p1_nash": p1, "p2_nash": p2,
        "p1_empirical": np.full(m, 1.0 / m), "p2_empirical": np.full(n, 1.0 / n),
    }


def process_output(out):
    """MCTS::Search::process_output (mcts.h:620-659) on a dict holding m, n, visit_matrix, value_matrix, iterations:
    adds nash_value / p1_nash / p2_nash (empirical matrix x 256 as integers, exact solve) and the empirical fields."""
    m, n = out["m"], out["n"]
    visits = np.asarray(out["visit_matrix"])[:m, :n].astype(np.int64)
    values = np.asarray(out["value_matrix"])[:m, :n].astype(np.float64)
    nn = np.where(visits == 0, 1, visits)
    p1, p2, nash = solve_matrix((values / nn * 256).astype(np.int64), 256)   # C++ double -> int truncation
    it = max(int(out["iterations"]), 1)
    out.update(nash_value=nash, p1_nash=p1, p2_nash=p2, empirical_value=float(values.sum() / it),
               p1_empirical=visits.sum(axis=1) / it, p2_empirical=visits.sum(axis=0) / it)
    return out


class Heap:
    """RuntimeSearch::Heap (util/search.h:17-32, search.cc:17-58) over oakgpu_heap: the tree of a search, kept between
    searches.  empty() is True until a search used it; update(i, j, obs) promotes the child reached by the PLAYED joint
    action (indices into the root's choice lists) and the 16-byte observation to the root, dropping the rest."""

    def __init__(self):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.oakgpu_heap_create(C.byref(h)))
        self.handle = h

    def empty(self):
        return bool(self.lib.oakgpu_heap_empty(self.handle))

    def nodes(self):
        return int(self.lib.oakgpu_heap_nodes(self.handle))

    def clear(self):
        self.lib.oakgpu_heap_clear(self.handle)

    def shard_violations(self):
        """Edges whose child sits in another table's arena (oakgpu_heap_check_shards): 0 in a consistent tree."""
        return int(self.lib.oakgpu_heap_check_shards(self.handle))

    def update(self, i, j, obs):
        o = np.ascontiguousarray(obs, dtype=np.uint8).reshape(16)
        return bool(self.lib.oakgpu_heap_update(self.handle, int(i), int(j), o.ctypes.data_as(C.c_void_p)))

    def root_stats(self, player):
        """(scores[k], priors[k], visits[k]) of the root's bandit of `player` (0 / 1); k = 0: the root is not initialised."""
        sc, pr, vi, k = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(9, np.uint32), C.c_uint8(0)
        _lib.check(self.lib.oakgpu_heap_root_stats(self.handle, int(player), sc.ctypes.data_as(C.c_void_p), pr.ctypes.data_as(C.c_void_p),
                                                   vi.ctypes.data_as(C.c_void_p), C.cast(C.byref(k), C.c_void_p)))
        return sc[:k.value].copy(), pr[:k.value].copy(), vi[:k.value].copy()

    def child_stats(self, i, j, obs, player):
        """root_stats of the child that update(i, j, obs) would promote (empty arrays: no such initialised child)."""
        o = np.ascontiguousarray(obs, dtype=np.uint8).reshape(16)
        sc, pr, vi, k = np.zeros(9, np.float32), np.zeros(9, np.float32), np.zeros(9, np.uint32), C.c_uint8(0)
        _lib.check(self.lib.oakgpu_heap_child_stats(self.handle, int(i), int(j), o.ctypes.data_as(C.c_void_p), int(player),
                                                    sc.ctypes.data_as(C.c_void_p), pr.ctypes.data_as(C.c_void_p), vi.ctypes.data_as(C.c_void_p),
                                                    C.cast(C.byref(k), C.c_void_p)))
        return sc[:k.value].copy(), pr[:k.value].copy(), vi[:k.value].copy()

    def close(self):
        if self.handle:
            self.lib.oakgpu_heap_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _output_dict(res):
    m, n = int(res.m), int(res.n)
    return {"m": m, "n": n, "p1_choices": np.array(res.p1_choices[:m], dtype=np.uint8),
            "p2_choices": np.array(res.p2_choices[:n], dtype=np.uint8),
            "visit_matrix": np.array(res.visit_matrix, dtype=np.int64).reshape(9, 9)[:m, :n].copy(),
            "value_matrix": np.array(res.value_matrix, dtype=np.float64).reshape(9, 9)[:m, :n].copy(),
            "iterations": int(res.iterations), "initial_value": float(res.initial_value), "nodes": int(res.nodes),
            "mean_depth": res.total_depth / max(int(res.iterations), 1), "duration_ms": res.duration_us / 1e3,
            # process_output ran in C++ (oakgpu_search_output: exact Nash of the empirical root matrix)
            "nash_value": float(res.nash_value), "p1_nash": np.array(res.p1_nash[:m]), "p2_nash": np.array(res.p2_nash[:n]),
            "empirical_value": float(res.empirical_value), "p1_empirical": np.array(res.p1_empirical[:m]),
            "p2_empirical": np.array(res.p2_empirical[:n]),
            "p1_logit": np.array(res.p1_logit[:m]), "p2_logit": np.array(res.p2_logit[:n]),
            "p1_prior": np.array(res.p1_prior[:m]), "p2_prior": np.array(res.p2_prior[:n]),
            "raw": res}          # the oakgpu_search_output itself: pass the dict back as `previous` to resume (mcts.h:153-155)


def tree_search_many(ctxs, battles, durations, results, seeds, iterations=1 << 16, batch=4096, c=2.0, bandit="ucb", evaluator="mc",
                     root_rolls=3, other_rolls=1, max_depth=100, alpha=0.05, heaps=None, threads_per_search=0):
    """n independent tree searches at once on one GPU (include/oakgpu.h: oakgpu_search_many): search i runs on ctxs[i] (a Context of its
    own each) from battles[i] with seeds[i]; heaps: None or one Heap per search.  Returns the list of output dicts -- each identical to
    tree_search(ctxs[i], battles[i], ..., seed=seeds[i]) run alone."""
    n = len(ctxs)
    battles = np.ascontiguousarray(battles, dtype=np.uint8).reshape(n, 384)
    durations = np.ascontiguousarray(durations, dtype=np.uint8).reshape(n, 8)
    results = np.ascontiguousarray(results, dtype=np.uint8).reshape(n)
    use_net = not isinstance(evaluator, str)
    prms = (_lib.SearchParams * n)()
    for i in range(n):
        prms[i] = _lib.SearchParams(iterations=int(iterations), batch=int(batch), ucb_c=float(c), bandit={"ucb": 0, "pucb": 1, "ucb1": 2, "exp3": 3, "pexp3": 4}[bandit],
                                    eval=1 if use_net else {"mc": 0, "poke-engine": 2}[evaluator], max_depth=int(max_depth), root_rolls=int(root_rolls),
                                    other_rolls=int(other_rolls), seed=int(seeds[i]), matrix_ucb=0, mucb_delay=0, mucb_minimum=0, mucb_c=0.0,
                                    exp3_alpha=float(alpha), duration_us=0)
    outs = (_lib.SearchOutput * n)()
    cp = (C.c_void_p * n)(*[c_.handle for c_ in ctxs])
    hp = (C.c_void_p * n)(*[h.handle for h in heaps]) if heaps is not None else None
    _lib.check(ctxs[0].lib.oakgpu_search_many(cp, evaluator.handle if use_net else None, hp, battles.ctypes.data_as(C.c_void_p),
                                              durations.ctypes.data_as(C.c_void_p), results.ctypes.data_as(C.c_void_p), prms, n, int(threads_per_search), outs))
    res = []
    for i in range(n):   # (_output_dict keeps a reference to its raw struct: give each its own copy)
        o = _lib.SearchOutput()
        C.memmove(C.byref(o), C.byref(outs[i]), C.sizeof(_lib.SearchOutput))
        res.append(_output_dict(o))
    return res


def tree_search(ctx, battle, durations, result, iterations=1 << 16, batch=4096, c=2.0, bandit="ucb", evaluator="mc",
                root_rolls=3, other_rolls=1, max_depth=100, seed=0x5EED, matrix_ucb=None, alpha=0.05, heap=None, previous=None,
                duration_us=0):
    """Tree search with batched leaves on the GPU (include/oakgpu.h: oakgpu_search_heap).  bandit: "ucb" | "pucb" | "ucb1" | "exp3" | "pexp3" (c = gamma for the Exp3 family, alpha its uniform mixing);
    evaluator: "mc", "poke-engine" (PokeEngine::Eval) or an oak_amd.engine.Network.  Defaults follow the reference's default_search{3, 1} damage-roll
    clamping (mcts.h:131).  matrix_ucb: None or (delay, minimum, c) -- MatrixUCBParams with interval = batch (the
    reference's "delay-interval-minimum-c" agent string, search.cc:216-231).  heap: None (fresh tree) or a Heap kept between
    searches; previous: None or the dict a previous tree_search of the SAME position returned -- matrices, iterations and
    duration accumulate like MCTS::Search::run's by-value Output (mcts.h:153-155, 231-247).  Returns a dict shaped like
    MCTS::Output (mcts.h:68-90)."""
    battle = np.ascontiguousarray(battle, dtype=np.uint8).reshape(384)
    durations = np.ascontiguousarray(durations, dtype=np.uint8).reshape(8)
    use_net = not isinstance(evaluator, str)
    prm = _lib.SearchParams(iterations=int(iterations), batch=int(batch), ucb_c=float(c), bandit={"ucb": 0, "pucb": 1, "ucb1": 2, "exp3": 3, "pexp3": 4}[bandit],
                            eval=1 if use_net else {"mc": 0, "poke-engine": 2}[evaluator], max_depth=int(max_depth), root_rolls=int(root_rolls),
                            other_rolls=int(other_rolls), seed=int(seed), matrix_ucb=1 if matrix_ucb else 0,
                            mucb_delay=int(matrix_ucb[0]) if matrix_ucb else 0, mucb_minimum=int(matrix_ucb[1]) if matrix_ucb else 0,
                            mucb_c=float(matrix_ucb[2]) if matrix_ucb else 0.0, exp3_alpha=float(alpha), duration_us=int(duration_us))
    res = _lib.SearchOutput()
    prev = C.byref(previous["raw"]) if previous is not None else None
    _lib.check(ctx.lib.oakgpu_search_heap(ctx.handle, evaluator.handle if use_net else None, heap.handle if heap is not None else None,
                                          battle.ctypes.data_as(C.c_void_p), durations.ctypes.data_as(C.c_void_p), int(result), C.byref(prm),
                                          prev, C.byref(res)))
    return _output_dict(res)
