"""GPU: Nash-at-root search (BASELINE config 5) fed by batched GPU updates / rollouts / leaf evals."""
import os

import numpy as np
import pytest

from oak_amd.parse import parse_battle, result_from_state
from oak_amd.search import root_matrix_search

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_single_action_positions_reproduce_search_test_values(gpu_ctx):
    # cpp/src/search-test.cc:80-108: wake-up probability 1 / (7 - K)
    for k, expected in ((3, 1 / 4), (5, 1 / 2)):
        b, d = parse_battle("starmie seismictoss 101hp slp%d | snorlax seismictoss 1hp" % k)
        out = root_matrix_search(gpu_ctx, b, d, result_from_state(b), replicas=16384, seed=k)
        assert out["m"] == 1 and out["n"] == 1
        assert abs(out["nash_value"] - expected) <= 0.03 and abs(out["empirical_value"] - expected) <= 0.03


def test_dominant_action_gets_all_the_nash_weight(gpu_ctx):
    # faster Starmie: Surf KOs the 1-hp Rhydon (value 1); Recover lets Rhydon KO it (value 0)
    b, d = parse_battle("starmie surf recover 1hp | rhydon earthquake 1hp")
    out = root_matrix_search(gpu_ctx, b, d, result_from_state(b), replicas=512)
    assert out["m"] == 2 and out["n"] == 1
    surf = [i for i, c in enumerate(out["p1_choices"]) if int(c) == ((1 << 2) | 1)][0]
    assert out["p1_nash"][surf] == 1.0 and out["nash_value"] == 1.0
    assert (out["value_matrix"] / out["visit_matrix"])[surf, 0] == 1.0


def test_full_position_with_network_evaluator(gpu_ctx):
    from oak_amd.engine import Network
    import oracle_lib as O
    b, d, p, r = O.make_random_ou_batch(1, seed0=31337)
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    for ev in ("mc", net):
        out = root_matrix_search(gpu_ctx, b[0], d[0], int(r[0]), replicas=64, evaluator=ev, seed=9)
        assert out["m"] == 9 and out["n"] == 9                      # 4 moves + 5 switches per side at turn 1
        assert abs(out["p1_nash"].sum() - 1) < 1e-9 and abs(out["p2_nash"].sum() - 1) < 1e-9
        assert 0.0 <= out["nash_value"] <= 1.0 and out["iterations"] == 81 * 64
        mean = out["value_matrix"] / out["visit_matrix"]
        # equilibrium property on the discretised matrix the solver saw
        M = np.floor(mean * 256)
        assert (M @ out["p2_nash"]).max() <= out["nash_value"] * 256 + 1e-6
        assert (out["p1_nash"] @ M).min() >= out["nash_value"] * 256 - 1e-6
    net.close()


# ---- tree search with batched leaves (oakgpu_search: host tree + bandits, GPU states / steps / leaves) ----------
def test_tree_search_single_action_positions(gpu_ctx):
    """search-test.cc:80-108 through the full tree search: one legal joint action, value = wake-up probability."""
    from oak_amd.search import tree_search
    for k, expected in ((3, 1 / 4), (5, 1 / 2)):
        b, d = parse_battle("starmie seismictoss 101hp slp%d | snorlax seismictoss 1hp" % k)
        out = tree_search(gpu_ctx, b, d, result_from_state(b), iterations=1 << 15, batch=4096, seed=k)
        assert out["m"] == 1 and out["n"] == 1 and out["iterations"] == 1 << 15
        assert int(out["visit_matrix"].sum()) == 1 << 15
        assert abs(out["empirical_value"] - expected) <= 0.03 and abs(out["nash_value"] - expected) <= 0.03


def test_tree_search_finds_the_dominant_action(gpu_ctx):
    from oak_amd.search import tree_search
    b, d = parse_battle("starmie surf recover 1hp | rhydon earthquake 1hp")
    out = tree_search(gpu_ctx, b, d, result_from_state(b), iterations=8192, batch=512, c=1.0)
    surf = [i for i, c in enumerate(out["p1_choices"]) if int(c) == ((1 << 2) | 1)][0]
    assert out["p1_nash"][surf] == 1.0 and out["nash_value"] == 1.0
    assert out["visit_matrix"][surf, 0] > 0.9 * out["iterations"]           # UCB concentrates on the winning move
    assert out["value_matrix"][surf, 0] == out["visit_matrix"][surf, 0]       # it always wins


def test_tree_search_invariants_and_determinism(gpu_ctx):
    """Full 9 x 9 root: bookkeeping invariants, same seed -> same tree, batch size changes the order not the sanity,
    and the tree search agrees with the one-ply matrix search on the root's value within sampling noise."""
    import oracle_lib as O
    from oak_amd.search import tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=424242)
    res = int(r[0])
    a = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=7)
    a2 = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=7)
    assert a["m"] == 9 and a["n"] == 9
    assert (a["visit_matrix"] == a2["visit_matrix"]).all() and (a["value_matrix"] == a2["value_matrix"]).all()
    assert int(a["visit_matrix"].sum()) == a["iterations"] == 1 << 14
    assert (a["value_matrix"] >= 0).all() and (a["value_matrix"] <= a["visit_matrix"] + 1e-9).all()
    assert a["nodes"] > 81 and 1.0 <= a["mean_depth"] <= 100.0
    assert abs(a["p1_nash"].sum() - 1) < 1e-9 and abs(a["p2_nash"].sum() - 1) < 1e-9 and 0 <= a["nash_value"] <= 1
    one = tree_search(gpu_ctx, b[0], d[0], res, iterations=2048, batch=1, seed=7)      # the reference's sequential order
    assert int(one["visit_matrix"].sum()) == 2048
    assert abs(one["empirical_value"] - a["empirical_value"]) < 0.1
    # unclamped rolls (39) are accepted too and give a similar value
    full = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=8, root_rolls=39, other_rolls=39)
    assert abs(full["empirical_value"] - a["empirical_value"]) < 0.1


def test_tree_search_with_network_and_pucb(gpu_ctx):
    import oracle_lib as O
    from oak_amd.engine import Network
    from oak_amd._lib import OakGpuError
    from oak_amd.search import tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=777)
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    for bandit in ("ucb", "pucb"):
        out = tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=4096, batch=512, bandit=bandit, evaluator=net, c=1.5)
        assert int(out["visit_matrix"].sum()) == 4096 and out["nodes"] > 81
        assert 0 <= out["empirical_value"] <= 1 and abs(out["p1_nash"].sum() - 1) < 1e-9
        if bandit == "pucb":
            assert 0 < out["initial_value"] < 1      # root value_policy_inference (mcts.h:196-209)
    with pytest.raises(OakGpuError):
        tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=64, batch=64, bandit="pucb", evaluator="mc")
    net.close()


def test_tree_search_with_poke_engine_eval(gpu_ctx):
    import oracle_lib as O
    from oak_amd.search import tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=99)
    out = tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=8192, batch=1024, evaluator="poke-engine", c=1.0)
    assert int(out["visit_matrix"].sum()) == 8192 and out["nodes"] > 81
    assert 0.2 < out["empirical_value"] < 0.8      # values are sigmoids of score differences from the root: near 0.5


def test_tree_search_matrix_ucb_root(gpu_ctx):
    """MatrixUCBParams (mcts.h:107-113,263-302,498-566): after `delay` iterations the root joint action is sampled from the
    Nash strategies of the UCB matrices; cells below `minimum` visits are forced first."""
    import oracle_lib as O
    from oak_amd.search import tree_search
    # dominant action: the sampling must concentrate on it and the value must come out exact
    b, d = parse_battle("starmie surf recover 1hp | rhydon earthquake 1hp")
    out = tree_search(gpu_ctx, b, d, result_from_state(b), iterations=8192, batch=512, c=1.0, matrix_ucb=(512, 8, 1.0))
    surf = [i for i, c in enumerate(out["p1_choices"]) if int(c) == ((1 << 2) | 1)][0]
    assert out["nash_value"] == 1.0 and out["visit_matrix"][surf, 0] > 0.8 * out["iterations"]
    assert (out["visit_matrix"] >= 8).all()                                  # the minimum-visits rule
    # full 9 x 9 root: every cell reaches the minimum, bookkeeping holds, value agrees with the plain UCB search
    bb, dd, pp, rr = O.make_random_ou_batch(1, seed0=424242)
    res = int(rr[0])
    plain = tree_search(gpu_ctx, bb[0], dd[0], res, iterations=1 << 15, batch=2048, seed=3)
    mu = tree_search(gpu_ctx, bb[0], dd[0], res, iterations=1 << 15, batch=2048, seed=3, matrix_ucb=(4096, 32, 0.5))
    assert int(mu["visit_matrix"].sum()) == 1 << 15 and (mu["visit_matrix"] >= 32).all()
    assert abs(mu["nash_value"] - plain["nash_value"]) < 0.12
    mu2 = tree_search(gpu_ctx, bb[0], dd[0], res, iterations=1 << 15, batch=2048, seed=3, matrix_ucb=(4096, 32, 0.5))
    assert (mu["visit_matrix"] == mu2["visit_matrix"]).all()               # reproducible


@pytest.mark.parametrize("bandit,c", [("ucb1", 2.0), ("exp3", 0.1)])
def test_tree_search_other_bandits(gpu_ctx, bandit, c):
    """UCB1::Bandit (ucb1.h) and Exp3::Bandit (exp3.h): the dominant action wins, bookkeeping holds, seeds reproduce."""
    import oracle_lib as O
    from oak_amd.search import tree_search
    b, d = parse_battle("starmie surf recover 1hp | rhydon earthquake 1hp")
    out = tree_search(gpu_ctx, b, d, result_from_state(b), iterations=8192, batch=256, c=c, bandit=bandit)
    surf = [i for i, ch in enumerate(out["p1_choices"]) if int(ch) == ((1 << 2) | 1)][0]
    assert out["nash_value"] == 1.0 and out["visit_matrix"][surf, 0] > 0.6 * out["iterations"]
    bb, dd, pp, rr = O.make_random_ou_batch(1, seed0=555)
    a1 = tree_search(gpu_ctx, bb[0], dd[0], int(rr[0]), iterations=1 << 14, batch=1024, c=c, bandit=bandit, seed=11)
    a2 = tree_search(gpu_ctx, bb[0], dd[0], int(rr[0]), iterations=1 << 14, batch=1024, c=c, bandit=bandit, seed=11)
    assert int(a1["visit_matrix"].sum()) == 1 << 14 and (a1["visit_matrix"] == a2["visit_matrix"]).all()
    assert a1["nodes"] > 81 and 0 <= a1["nash_value"] <= 1


def test_tree_search_pexp3_with_network(gpu_ctx):
    import oracle_lib as O
    from oak_amd.engine import Network
    from oak_amd.search import tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=777)
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    out = tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=4096, batch=512, bandit="pexp3", evaluator=net, c=0.1)
    assert int(out["visit_matrix"].sum()) == 4096 and out["nodes"] > 81 and 0 < out["initial_value"] < 1
    net.close()


SEARCH_TEST_POSITIONS = (  # cpp/src/search-test.cc:50-109: (position, expected value, tolerance)
    ("starmie seismictoss 1hp (conf:5) | snorlax bodyslam 1hp", 1.0, 0.0),
    ("starmie seismictoss 1hp (conf:4) | snorlax bodyslam 1hp", .5 + .5 / 2, 0.03),
    ("starmie seismictoss 1hp (conf:3) | snorlax bodyslam 1hp", .33 + .66 / 2, 0.03),
    ("starmie seismictoss 1hp (conf:2) | snorlax bodyslam 1hp", .25 + .75 / 2, 0.03),
    ("starmie seismictoss 1hp (conf:1) | snorlax bodyslam 1hp", .5, 0.03),
    ("starmie seismictoss 1hp slp6 | snorlax seismictoss 1hp", 0.0, 0.0),
    ("starmie seismictoss 101hp slp0 | snorlax seismictoss 1hp", 1.0 / 7, 0.03),
    ("starmie seismictoss 101hp slp1 | snorlax seismictoss 1hp", 1.0 / 6, 0.03),
    ("starmie seismictoss 101hp slp2 | snorlax seismictoss 1hp", 1.0 / 5, 0.03),
    ("starmie seismictoss 101hp slp3 | snorlax seismictoss 1hp", 1.0 / 4, 0.03),
    ("starmie seismictoss 101hp slp4 | snorlax seismictoss 1hp", 1.0 / 3, 0.03),
    ("starmie seismictoss 101hp slp5 | snorlax seismictoss 1hp", 1.0 / 2, 0.03),
    ("starmie seismictoss 101hp slp6 | snorlax seismictoss 1hp", 1.0, 0.0),
)


def test_search_test_cc_all_13_positions_through_the_tree_search(gpu_ctx):
    """The reference's only runtime test, configured as the reference configures it (search-test.cc:27-31): bandit
    "exp3-1.0-0.1", Monte-Carlo leaves, 2^20 iterations per position, abs(empirical_value - expected) <= 0.03 (exactly
    0 for the three deterministic positions) -- through oakgpu_search (host tree + Exp3 bandits, GPU states and leaves)."""
    from oak_amd.search import tree_search
    for k, (position, expected, tol) in enumerate(SEARCH_TEST_POSITIONS):
        b, d = parse_battle(position)
        out = tree_search(gpu_ctx, b, d, result_from_state(b), iterations=1 << 20, batch=16384, bandit="exp3", c=1.0, alpha=0.1,
                          seed=0xC0FFEE + k)
        assert out["iterations"] == 1 << 20 and out["m"] == 1 and out["n"] == 1
        assert abs(out["empirical_value"] - expected) <= tol, (position, out["empirical_value"], expected)
        assert abs(out["nash_value"] - expected) <= tol + 1 / 256   # process_output's x256 integer matrix


def test_tree_search_output_carries_the_exact_nash_solution(gpu_ctx):
    """oakgpu_search_output.nash_* (process_output in C++, mcts.h:620-659) = oakgpu_solve_matrix of the empirical root
    matrix x 256 truncated to integers, for all five bandits on a 9 x 9 root."""
    import oracle_lib as O
    from oak_amd.engine import Network
    from oak_amd.search import solve_matrix, tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=777)
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    for bandit, ev, c in (("ucb", "mc", 2.0), ("ucb1", "mc", 2.0), ("exp3", "mc", 0.3), ("pucb", net, 1.5), ("pexp3", net, 0.5)):
        out = tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=1 << 13, batch=1024, bandit=bandit, c=c, evaluator=ev, seed=3)
        v, n = out["value_matrix"], out["visit_matrix"]
        M = (v / np.where(n == 0, 1, n) * 256).astype(np.int64)
        p1, p2, nv = solve_matrix(M, 256)
        assert np.allclose(p1, out["p1_nash"], atol=1e-12) and np.allclose(p2, out["p2_nash"], atol=1e-12)
        assert abs(nv - out["nash_value"]) <= 1e-12 and abs(out["p1_nash"].sum() - 1) < 1e-9
        assert (M @ p2).max() <= nv * 256 + 1e-6 and (p1 @ M).min() >= nv * 256 - 1e-6    # equilibrium of the matrix solved
    net.close()


# ---- RuntimeSearch::Heap + the resumable Output of Search::run (mcts.h:153-155, 231-247; search.cc:27-52) -------------
def test_resumed_output_accumulates_like_search_run(gpu_ctx):
    """Two searches of 2^14 through one Heap with the first Output passed back in: iterations == 2^15, matrices == the sum
    of what each search contributed, duration adds up -- `Output output = {}` is by value and added to (mcts.h:153-155)."""
    import oracle_lib as O
    from oak_amd.search import Heap, tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=424242)
    res = int(r[0])
    h1, h2 = Heap(), Heap()
    assert h1.empty() and h1.nodes() == 0
    o1 = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=1, heap=h1)
    assert not h1.empty() and h1.nodes() == o1["nodes"]
    o2 = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=2, heap=h1, previous=o1)
    # the same two searches through a second heap, the second one NOT resumed: its output is only its own contribution
    p1 = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=1, heap=h2)
    p2 = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=2, heap=h2)
    assert (p1["visit_matrix"] == o1["visit_matrix"]).all() and (p1["value_matrix"] == o1["value_matrix"]).all()   # same seed, same tree
    assert o2["iterations"] == 1 << 15 and p2["iterations"] == 1 << 14
    assert (o2["visit_matrix"] == o1["visit_matrix"] + p2["visit_matrix"]).all()
    assert np.allclose(o2["value_matrix"], o1["value_matrix"] + p2["value_matrix"], rtol=0, atol=1e-6)
    assert int(o2["visit_matrix"].sum()) == 1 << 15
    assert abs(o2["empirical_value"] - o2["value_matrix"].sum() / (1 << 15)) < 1e-12          # process_output over the sums
    assert np.allclose(o2["p1_empirical"], o2["visit_matrix"].sum(axis=1) / float(1 << 15))
    assert o2["duration_ms"] > o1["duration_ms"] > 0                                           # output.duration += ...
    assert o2["nodes"] > o1["nodes"]                                                           # the tree kept growing
    # the second search really used the first one's statistics: a fresh-tree search with seed 2 is a different search
    fresh = tree_search(gpu_ctx, b[0], d[0], res, iterations=1 << 14, batch=2048, seed=2)
    assert not (fresh["visit_matrix"] == p2["visit_matrix"]).all()
    h1.close(); h2.close()


def test_heap_update_promotes_the_played_child(gpu_ctx):
    """Heap::update(i, j, obs) (search.cc:27-52): the child's bandit statistics become the root's, bit for bit; the next
    search continues from them; an edge the search never took leaves an uninitialised root; a heap holds one bandit type."""
    import oracle_lib as O
    from oak_amd._lib import OakGpuError
    from oak_amd.search import Heap, tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=31337)
    battle, dur, res = b[0].copy(), d[0].copy(), int(r[0])
    h = Heap()
    o1 = tree_search(gpu_ctx, battle, dur, res, iterations=1 << 15, batch=2048, seed=5, heap=h)
    i, j = np.unravel_index(np.argmax(o1["visit_matrix"]), o1["visit_matrix"].shape)
    # play the most visited joint action for real; several seeds of the battle's own RNG until the observation is one the tree holds
    for attempt in range(64):
        nb, nd = battle.copy().reshape(1, 384), dur.copy().reshape(1, 8)
        nb[0, 376:384] = np.frombuffer(np.uint64(0x9E3779B97F4A7C15 * (attempt + 1) & 0xFFFFFFFFFFFFFFFF).tobytes(), dtype=np.uint8)
        nres, actions = gpu_ctx.update(nb, np.array([o1["p1_choices"][i]], np.uint8), np.array([o1["p2_choices"][j]], np.uint8), nd)
        before = [h.child_stats(i, j, actions[0], pl) for pl in (0, 1)]
        if len(before[0][0]):
            break
    assert len(before[0][0]) > 0, "no observation of the most visited joint action is in the tree"
    nodes_before = h.nodes()
    assert h.update(i, j, actions[0]) is True
    after = [h.root_stats(pl) for pl in (0, 1)]
    for pl in (0, 1):
        for x, y in zip(before[pl], after[pl]):
            assert x.tobytes() == y.tobytes()                        # scores, priors, visits: the old child's, exactly
    assert 0 < h.nodes() < nodes_before                              # the rest of the tree is dropped
    assert h.shard_violations() == 0                                 # every kept edge's child was re-homed to the edge's new table
    kept = h.nodes()
    if (int(nres[0]) & 15) == 0:
        o2 = tree_search(gpu_ctx, nb[0], nd[0], int(nres[0]), iterations=1 << 13, batch=1024, seed=6, heap=h)
        assert o2["iterations"] == 1 << 13 and o2["nodes"] > kept and h.shard_violations() == 0
        visits_root = h.root_stats(0)[2]
        assert int(visits_root.sum()) == int(after[0][2].sum()) + (1 << 13)     # UCB: every iteration is one more visit at the root
        # an observation that cannot exist: nothing is kept, the heap holds an uninitialised node (not monostate)
        assert h.update(0, 0, np.full(16, 0xEE, np.uint8)) is False
        assert not h.empty() and h.nodes() == 0 and len(h.root_stats(0)[0]) == 0
        o3 = tree_search(gpu_ctx, nb[0], nd[0], int(nres[0]), iterations=4096, batch=1024, seed=7, heap=h)
        assert o3["iterations"] == 4096
        with pytest.raises(OakGpuError, match="Bad Heap access"):
            tree_search(gpu_ctx, nb[0], nd[0], int(nres[0]), iterations=1024, batch=256, bandit="exp3", c=0.1, heap=h)
    h.close()


def test_a_heap_searched_from_another_position_is_refused_with_its_own_code(gpu_ctx):
    """A heap whose root was initialised for one position and is then searched from a position with other legal-choice counts (the
    caller forgot Heap::update, or a --keep-node game reached a position its kept child was not expanded for): refused with
    OAKGPU_E_ROOT_MISMATCH (-2) -- the code oakgpu_selfplay_game recovers from by starting the position's tree afresh -- not with a
    message that has to be parsed."""
    from oak_amd._lib import OakGpuError
    from oak_amd.search import Heap, tree_search
    full, d9 = parse_battle("starmie surf recover thunderwave psychic; alakazam psychic; chansey icebeam | snorlax bodyslam rest; tauros bodyslam; exeggutor psychic")
    single, d1 = parse_battle("starmie seismictoss 101hp slp3 | snorlax seismictoss 1hp")
    h = Heap()
    a = tree_search(gpu_ctx, full, d9, result_from_state(full), iterations=2048, batch=512, seed=1, heap=h)
    assert a["m"] > 1
    with pytest.raises(OakGpuError, match=r"other action counts.*code -2"):
        tree_search(gpu_ctx, single, d1, result_from_state(single), iterations=512, batch=128, seed=2, heap=h)
    b = tree_search(gpu_ctx, full, d9, result_from_state(full), iterations=2048, batch=512, seed=3, heap=h)     # the heap is still good for ITS position
    assert b["iterations"] == 2048 and b["nodes"] >= a["nodes"]
    h.close()


def test_time_budget_on_a_cold_context_runs_at_least_one_batch():
    """Round-2 advice: with the clock started before the set-up a 1 ms budget on a fresh context ran zero batches and
    returned NaN strategies.  The clock now covers the iteration loop only and a time budget always runs once
    (`while (elapsed < duration)` starts at elapsed = 0, mcts.h:219-226)."""
    from oak_amd.engine import Context
    from oak_amd.search import tree_search
    import oracle_lib as O
    b, d, p, r = O.make_random_ou_batch(1, seed0=2024)
    ctx = Context(0)
    out = tree_search(ctx, b[0], d[0], int(r[0]), iterations=0, batch=1024, duration_us=1000, seed=1)
    assert out["iterations"] >= 1024 and out["iterations"] % 1024 == 0
    assert np.isfinite(out["p1_empirical"]).all() and abs(out["p1_empirical"].sum() - 1) < 1e-6
    assert np.isfinite(out["empirical_value"]) and 0 < out["duration_ms"] < 2000
    # a second search on the now warm context: the budget is honoured within a batch or two
    out2 = tree_search(ctx, b[0], d[0], int(r[0]), iterations=0, batch=1024, duration_us=20000, seed=2)
    assert 20 <= out2["duration_ms"] < 200 and out2["iterations"] >= 1024
    ctx.close()


def test_contextual_root_priors_and_zero_budget(gpu_ctx):
    """Output::Side::logit / prior (mcts.h:196-209): the root's legal logits and softmax(logits) for PUCB / PExp3; budget 0
    runs no iteration and still fills them (what cpp_inference reads, pyoak.cc:331-392)."""
    import oracle_lib as O
    from oak_amd.engine import Network
    from oak_amd.search import tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=777)
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    out = tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=0, batch=1, bandit="pucb", evaluator=net, c=1.0)
    assert out["iterations"] == 0 and int(out["visit_matrix"].sum()) == 0
    for side in ("p1", "p2"):
        lg, pr = out[side + "_logit"], out[side + "_prior"]
        e = np.exp(lg.astype(np.float32))
        assert np.allclose(pr, e / e.sum(dtype=np.float32), atol=1e-6) and abs(pr.sum() - 1) < 1e-6
    assert abs(out["initial_value"] - float(net.value_inference(b[:1], d[:1])[0])) <= 1e-6
    # a non-contextual bandit leaves them at zero, like the reference's value-initialised Output
    plain = tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=1024, batch=256, evaluator=net)
    assert not plain["p1_prior"].any() and not plain["p1_logit"].any()
    net.close()


def test_host_thread_count_does_not_change_the_search(gpu_ctx):
    """The tree is cut into 16 shards served by 1, 2, 4, 8 or 16 host threads (search_host.hip): any count walks the same tree."""
    import oracle_lib as O
    from oak_amd.search import tree_search
    b, d, p, r = O.make_random_ou_batch(1, seed0=999)
    outs = []
    old = os.environ.get("OAKGPU_SEARCH_THREADS")
    try:
        for w in ("1", "2", "4", "8", "16"):
            os.environ["OAKGPU_SEARCH_THREADS"] = w
            outs.append(tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=1 << 15, batch=4096, seed=13, bandit="exp3", c=0.3))
            outs.append(tree_search(gpu_ctx, b[0], d[0], int(r[0]), iterations=1 << 15, batch=4096, seed=13))
    finally:
        if old is None:
            os.environ.pop("OAKGPU_SEARCH_THREADS", None)
        else:
            os.environ["OAKGPU_SEARCH_THREADS"] = old
    for k in (2, 4, 6, 8):
        for base in (0, 1):
            assert (outs[k + base]["visit_matrix"] == outs[base]["visit_matrix"]).all()
            assert outs[k + base]["value_matrix"].tobytes() == outs[base]["value_matrix"].tobytes()
            assert outs[k + base]["nodes"] == outs[base]["nodes"]


def test_tutorial_known_answer_snorlax_vs_starmie(gpu_ctx):
    """The reference's other published known answer (TUTORIAL.md:46-70): `snorlax bodyslam rest | starmie psychic thunderwave
    recover`, agent ucb-1.0, Monte-Carlo leaves, 8 s (847,615 iterations there) => Value 0.448; the two heavily visited
    cells BodySlam x ThunderWave 0.443 (813,826 visits) and BodySlam x Psychic 0.602 (23,811).  Exercises paralysis, Rest,
    Body Slam's secondary, Psychic's special drop, speed order, crits and damage rolls -- far more of the engine than the 13
    sleep / confusion positions.  Tolerances: 0.03 on the value (the reference's own test tolerance), 0.05 on the cells."""
    from oak_amd import _lib
    import ctypes as C
    b, d = parse_battle("snorlax bodyslam rest | starmie psychic thunderwave recover")
    res = result_from_state(b)
    out = _lib.SearchOutput()
    agent = _lib.Agent(budget=b"1048576", bandit=b"ucb-1.0", eval=b"mc", matrix_ucb=b"", discrete=0, table=0)
    _lib.check(gpu_ctx.lib.oakgpu_search_agent(gpu_ctx.handle, b.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), int(res),
                                               C.byref(agent), 1024, 0x7157, C.byref(out)))
    from oak_amd.search import _output_dict
    o = _output_dict(out)
    assert o["m"] == 2 and o["n"] == 3 and o["iterations"] == 1 << 20
    names1 = {(1 << 2) | 1: "bodyslam", (2 << 2) | 1: "rest"}
    names2 = {(1 << 2) | 1: "psychic", (2 << 2) | 1: "thunderwave", (3 << 2) | 1: "recover"}
    r1 = {names1[int(c)]: k for k, c in enumerate(o["p1_choices"])}
    r2 = {names2[int(c)]: k for k, c in enumerate(o["p2_choices"])}
    mean = o["value_matrix"] / np.maximum(o["visit_matrix"], 1)
    report = {"value": o["empirical_value"], "bs_tw": mean[r1["bodyslam"], r2["thunderwave"]], "bs_psy": mean[r1["bodyslam"], r2["psychic"]],
              "visits": o["visit_matrix"].tolist(), "p1_emp": o["p1_empirical"].tolist(), "p2_emp": o["p2_empirical"].tolist()}
    print("TUTORIAL known answer:", report)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    import json
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "tutorial_known_answer.json"), "w"), indent=1)
    assert abs(o["empirical_value"] - 0.448) <= 0.03, report
    assert abs(mean[r1["bodyslam"], r2["thunderwave"]] - 0.443) <= 0.05, report
    assert abs(mean[r1["bodyslam"], r2["psychic"]] - 0.602) <= 0.05, report
    assert o["p1_empirical"][r1["bodyslam"]] > 0.9 and o["p2_empirical"][r2["thunderwave"]] > 0.8, report


def test_concurrent_searches_equal_the_searches_run_alone(gpu_ctx):
    """oakgpu_search_many (several roots per GPU at once): six independent searches on six contexts, their tree walks
    on two host threads each, Monte-Carlo leaves and -- a second round -- network leaves with PUCB priors: every output (visit and
    value matrices, node counts, Nash value) equals the same search run alone on the primary context, and so do the heaps' roots."""
    import oracle_lib as O
    from oak_amd.engine import Context, Network
    from oak_amd.search import Heap, tree_search, tree_search_many
    n = 6
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x51DE5)
    ctxs = [Context(0) for _ in range(n)]
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    try:
        for kw in (dict(evaluator="mc", bandit="ucb", c=2.0), dict(evaluator=net, bandit="pucb", c=1.0)):
            seeds = [1000 + 17 * i for i in range(n)]
            heaps = [Heap() for _ in range(n)]
            many = tree_search_many(ctxs, b, d, r, seeds, iterations=1 << 13, batch=1024, heaps=heaps, threads_per_search=2, **kw)
            for i in range(n):
                h = Heap()
                alone = tree_search(gpu_ctx, b[i], d[i], int(r[i]), iterations=1 << 13, batch=1024, seed=seeds[i], heap=h, **kw)
                for key in ("visit_matrix", "value_matrix"):
                    assert (many[i][key] == alone[key]).all(), (i, key)
                assert many[i]["nodes"] == alone["nodes"] and many[i]["iterations"] == alone["iterations"] == 1 << 13
                assert many[i]["nash_value"] == alone["nash_value"] and many[i]["mean_depth"] == alone["mean_depth"]
                for pl in (0, 1):
                    for x, y in zip(heaps[i].root_stats(pl), h.root_stats(pl)):
                        assert x.tobytes() == y.tobytes()
                h.close()
            for h in heaps:
                h.close()
    finally:
        net.close()
        for c in ctxs:
            c.close()


@pytest.mark.gpu
def test_tutorial_vs_known_answer_small_sample():
    """TUTORIAL.md:99-104: `vs --budget=4096 --bandit=ucb-1.0 --policy-mode=x --p1-eval=fp --p2-eval=mc` scores 186-1-31 for the PokeEngine
    agent over the 16 sample teams.  tools/tutorial_stats.py rebuilds vs.cc's loop on the GPU path (218 games: 179-0-39 at 64 descents per
    batch, profiles/r04_tutorial_stats.json); here a fixed-seed sample of 16 games through the same loop: the agent with the static
    evaluation must still win clearly (the run is deterministic for a given build; 10 of 16 is 1.5 sigma under the 218-game rate)."""
    import json
    import subprocess
    import sys
    env = dict(os.environ, SEED="20261004", CONC="16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "tutorial_stats.py"), "vs", "8", "256"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert rec["games"] == 16 and rec["W"] + rec["D"] + rec["L"] == 16
    assert rec["W"] + 0.5 * rec["D"] >= 10, rec
    assert 30 <= rec["mean_updates"] <= 200, rec


# ---- BASELINE configs[4] at bench.py's size (search_workload: 8 roots at once, 2^18 iterations, 768-256-256-256-1 net) --------------
def _nash_certificate(out, tol=1e-9):
    """The minimax certificate of MCTS::Search::process_output's solve (mcts.h:620-659): the matrix it solved is the empirical root
    matrix x 256 truncated to integers; (p1, p2, v) is an equilibrium of it iff no pure reply beats v -- checked in exact rationals from
    the floats the output carries (the solver's strategies are k / denominator, so the tolerance only has to cover their float form)."""
    from fractions import Fraction
    v, n = out["value_matrix"], out["visit_matrix"]
    M = (v / np.where(n == 0, 1, n) * 256).astype(np.int64)
    p1 = [Fraction(float(x)) for x in out["p1_nash"]]
    p2 = [Fraction(float(x)) for x in out["p2_nash"]]
    assert all(x >= 0 for x in p1) and all(x >= 0 for x in p2)
    assert abs(sum(p1) - 1) < tol and abs(sum(p2) - 1) < tol
    val = Fraction(float(out["nash_value"])) * 256
    rows = [sum(int(M[i, j]) * p2[j] for j in range(M.shape[1])) for i in range(M.shape[0])]     # P1's pure replies to p2
    cols = [sum(int(M[i, j]) * p1[i] for i in range(M.shape[0])) for j in range(M.shape[1])]     # P2's pure replies to p1
    assert max(rows) <= val + tol * 256 and min(cols) >= val - tol * 256, (float(max(rows)), float(min(cols)), float(val))
    return M


def test_full_size_config5_search_properties(gpu_ctx):
    """BASELINE configs[4] at the size bench.py times it (search_workload / the `config5` record): oakgpu_search_many, 8 random OU turn-1
    roots searched AT ONCE, 2^18 iterations each in batches of 16,384 descents, joint UCB (c = 2), the 768-256-256-256-1 network
    (tests/golden/net_256.battle.net, the reference torch mirror's file) as leaf evaluator, exact Nash of the root matrix.  No oracle
    exists for a search (the reference's is seeded from std::random_device, search-test.cc:20-21), so everything is held by
    properties (MCTS::Search::run, mcts.h:154-248; process_output, mcts.h:498-566, 620-659):
      * bookkeeping: every iteration is one root visit (visit sum = iterations = 2^18), 0 <= value sum <= visits per cell, every legal
        joint action is visited, nodes <= iterations + 1, the tree is deeper than one ply;
      * the Nash solution is an exact equilibrium of the matrix process_output solved (rational best-response certificate), both
        strategies are distributions, the value is inside the matrix's range;
      * same seeds -> the same eight outputs, byte for byte (the many-roots schedule is deterministic);
      * each of the eight equals the search run ALONE on another context with all host threads (visit / value matrices, nodes, depth,
        Nash value): concurrency changes nothing;
      * the empirical value and strategies are the matrices' own sums; eight different positions give eight different values."""
    import oracle_lib as O
    from oak_amd.engine import Context, Network
    from oak_amd.search import tree_search, tree_search_many
    R, iters, batch = 8, 1 << 18, 16384
    b, d, p, r = O.make_random_ou_batch(R, seed0=0x0A4B00000000 + 4096)      # bench.py's roots (SEED0 + 4096)
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_256.battle.net"))
    assert net.shape()[:3] == (768, 256, 256)
    ctxs = [Context(0) for _ in range(R)]
    try:
        seeds = list(range(R))
        many = tree_search_many(ctxs, b, d, r, seeds, iterations=iters, batch=batch, evaluator=net, threads_per_search=2)
        again = tree_search_many(ctxs, b, d, r, seeds, iterations=iters, batch=batch, evaluator=net, threads_per_search=2)
        values = []
        for i in range(R):
            o = many[i]
            assert o["iterations"] == iters and int(o["visit_matrix"].sum()) == iters
            assert o["m"] >= 1 and o["n"] >= 1 and o["visit_matrix"].shape == (o["m"], o["n"])
            assert (o["visit_matrix"] > 0).all()                                   # 2^18 iterations over <= 81 cells: none unvisited
            assert (o["value_matrix"] >= 0).all() and (o["value_matrix"] <= o["visit_matrix"] + 1e-6).all()
            assert 81 < o["nodes"] <= iters + 1 and 1.0 < o["mean_depth"] <= 100.0
            M = _nash_certificate(o)
            assert M.min() / 256 - 1e-9 <= o["nash_value"] <= M.max() / 256 + 1e-9
            emp = o["value_matrix"].sum() / iters
            assert abs(emp - o["empirical_value"]) < 1e-9 and 0.0 < emp < 1.0
            assert abs(o["p1_empirical"].sum() - 1) < 1e-9 and abs(o["p2_empirical"].sum() - 1) < 1e-9
            assert np.allclose(o["p1_empirical"], o["visit_matrix"].sum(axis=1) / iters, atol=1e-12)
            assert np.allclose(o["p2_empirical"], o["visit_matrix"].sum(axis=0) / iters, atol=1e-12)
            values.append(emp)
            for key in ("visit_matrix", "value_matrix", "p1_nash", "p2_nash"):
                assert (o[key] == again[i][key]).all(), (i, key)
            assert o["nodes"] == again[i]["nodes"] and o["nash_value"] == again[i]["nash_value"]
            alone = tree_search(gpu_ctx, b[i], d[i], int(r[i]), iterations=iters, batch=batch, seed=seeds[i], evaluator=net)
            for key in ("visit_matrix", "value_matrix", "p1_nash", "p2_nash"):
                assert (o[key] == alone[key]).all(), (i, key)
            assert o["nodes"] == alone["nodes"] and o["nash_value"] == alone["nash_value"] and o["mean_depth"] == alone["mean_depth"]
        assert len(set(round(v, 6) for v in values)) == R                          # eight different positions, eight different answers
    finally:
        net.close()
        for c in ctxs:
            c.close()
