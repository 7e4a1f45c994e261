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
