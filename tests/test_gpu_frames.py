"""GPU: self-play games on the GPU path -> `.battle.data` records -> the reference's replay self-check
(cpp/include/py/battle/frames.h:52-67): replaying the stored choices from the stored battle reproduces the stored result."""
import numpy as np
import pytest

import oracle_lib as O
from oak_amd.frames import read_frames, selfplay_game
from test_oracle_goldens import benchmark_teams

pytestmark = pytest.mark.gpu


def _replay(ctx, game):
    """frames.h:52-67 twice over: through the GPU update (the product) and through the CPU oracle (the checker)."""
    gb, gd = game["battle"].reshape(1, 384).copy(), np.zeros((1, 8), dtype=np.uint8)
    ob, opt = game["battle"].copy(), O.Options()
    gres = ores = O.LIB.oracle_result_from_state(O.ptr(ob))
    for u in game["updates"]:
        c1s, n1 = ctx.choices(gb, np.array([gres], dtype=np.uint8), 0)
        c2s, n2 = ctx.choices(gb, np.array([gres], dtype=np.uint8), 1)
        assert (n1[0], n2[0]) == (u["m"], u["n"])                       # the frame's m / n are the position's legal choices
        assert u["c1"] in c1s[0, :n1[0]] and u["c2"] in c2s[0, :n2[0]]
        r, _ = ctx.update(gb, [u["c1"]], [u["c2"]], gd, want_actions=False)
        gres = int(r[0])
        opt.set()
        ores = O.update(ob, u["c1"], u["c2"], opt)
        assert gres == ores and (gb[0] == ob).all()
    return gres


def test_selfplay_records_replay_to_their_stored_result(gpu_ctx):
    teams = np.array(benchmark_teams(), dtype=np.uint8)
    blob, meta = b"", []
    for g, (bandit, mode, ev) in enumerate((("ucb", "e", "mc"), ("exp3", "n", "mc"), ("ucb", "e0.9-x0.1", "poke-engine"))):
        rec, frames, result = selfplay_game(gpu_ctx, teams, battle_seed=1000 + g, iterations=512, batch=256, bandit=bandit,
                                            c=2.0 if bandit == "ucb" else 0.3, evaluator=ev, policy_mode=mode, seed=g + 1)
        assert frames >= 5 and (result & 15) in (1, 2, 3)
        blob += rec
        meta.append((frames, result))
    games = read_frames(blob)
    assert [(len(g["updates"]), g["result"]) for g in games] == meta
    for g in games:
        assert g["battle"][368] == 1                                    # stored after the opening update: turn 1
        assert _replay(gpu_ctx, g) == g["result"]                       # assert(result == compressed_frames.result), frames.h:67
        for u in g["updates"]:
            assert u["iterations"] == 512 and 0.0 <= u["empirical_value"] <= 1.0
            assert abs(u["p1_empirical"].sum() - 1) < 9 / 65535 + 1e-9 and abs(u["p1_nash"].sum() - 1) < 9 / 65535 + 1e-9


def test_selfplay_with_the_network_evaluator_and_length_cap(gpu_ctx):
    import os
    from oak_amd.engine import Network
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    net = Network(gpu_ctx, path=os.path.join(root, "tests", "golden", "net_default.battle.net"))
    teams = np.array(benchmark_teams(), dtype=np.uint8)
    rec, frames, result = selfplay_game(gpu_ctx, teams, battle_seed=77, iterations=256, batch=256, bandit="pucb", c=1.5, evaluator=net, seed=9)
    (game,) = read_frames(rec)
    assert len(game["updates"]) == frames and _replay(gpu_ctx, game) == result
    # 'p': the contextual bandit's prior mixed in (policy.h:37-46, the reference's fall-through into 'e')
    rec, frames, result = selfplay_game(gpu_ctx, teams, battle_seed=78, iterations=256, batch=256, bandit="pucb", c=1.5, evaluator=net,
                                        policy_mode="p0.5-x0.5", seed=10)
    (game,) = read_frames(rec)
    assert len(game["updates"]) == frames and _replay(gpu_ctx, game) == result
    with pytest.raises(RuntimeError, match="policy mode"):
        selfplay_game(gpu_ctx, teams, battle_seed=77, iterations=64, batch=64, policy_mode="b", seed=9)
    with pytest.raises(RuntimeError, match="max battle length"):
        selfplay_game(gpu_ctx, teams, battle_seed=77, iterations=64, batch=64, max_battle_length=3, seed=9)
    with pytest.raises(RuntimeError, match="policy mode"):
        selfplay_game(gpu_ctx, teams, battle_seed=77, iterations=64, batch=64, policy_mode="q", seed=9)
    net.close()


def test_selfplay_keep_node_searches_on_in_the_played_subtree(gpu_ctx):
    """generate's --keep-node (generate.cc:324-333): one heap for the whole game, Heap::update(p1_index, p2_index, obs) after
    every turn.  The record must replay exactly like any other, most updates find their child (the played action is the
    most searched one), and the frames still carry `iterations` per turn (each turn's output starts from zero)."""
    teams = np.array(benchmark_teams(), dtype=np.uint8)
    stats = {}
    rec, frames, result = selfplay_game(gpu_ctx, teams, battle_seed=4242, iterations=2048, batch=512, bandit="ucb", c=2.0, evaluator="mc",
                                        policy_mode="e", seed=5, keep_node=True, stats=stats)
    (game,) = read_frames(rec)
    assert len(game["updates"]) == frames and _replay(gpu_ctx, game) == result
    assert all(u["iterations"] == 2048 for u in game["updates"])
    assert 0 < stats["nodes_kept"] <= frames and stats["nodes_kept"] >= frames // 4


def test_concurrent_selfplay_games_are_the_games_played_alone(gpu_ctx):
    """oakgpu_selfplay_games (the generator's worker pool on one GPU, generate.cc:527-536): four games at once on four contexts, two
    host threads per game's tree walks, with and without --keep-node -- every record is byte for byte the record of the same game
    played alone, and replays to its stored result."""
    from oak_amd.engine import Context
    from oak_amd.frames import selfplay_games
    teams = np.array(benchmark_teams(), dtype=np.uint8)
    n = 4
    ctxs = [Context(0) for _ in range(n)]
    try:
        for keep in (False, True):
            kw = dict(iterations=512, batch=256, bandit="ucb", c=2.0, evaluator="mc", policy_mode="e", keep_node=keep)
            many = selfplay_games(ctxs, np.stack([teams] * n), [2000 + g for g in range(n)], [g + 1 for g in range(n)], threads_per_game=2, **kw)
            for g in range(n):
                alone = selfplay_game(gpu_ctx, teams, battle_seed=2000 + g, seed=g + 1, **kw)
                assert many[g] == alone, (keep, g, many[g][1:], alone[1:])
            game = read_frames(many[0][0])[0]
            assert _replay(gpu_ctx, game) == game["result"] == many[0][2]
    finally:
        for c in ctxs:
            c.close()
