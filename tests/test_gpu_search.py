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
