"""Manual GPU tool: oakgpu_selfplay_games vs oakgpu_selfplay_game, by number of games and host threads per game."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oak_amd.engine import Context
from oak_amd.frames import selfplay_game, selfplay_games
from test_oracle_goldens import benchmark_teams

teams = np.array(benchmark_teams(), dtype=np.uint8)
main = Context(0)
for keep in (False, True):
    kw = dict(iterations=512, batch=256, bandit="ucb", c=2.0, evaluator="mc", policy_mode="e", keep_node=keep)
    alone = [selfplay_game(main, teams, battle_seed=2000 + g, seed=g + 1, **kw) for g in range(4)]
    for n, tpg in ((1, 8), (1, 2), (1, 1), (4, 2), (4, 1)):
        ctxs = [Context(0) for _ in range(n)]
        try:
            many = selfplay_games(ctxs, np.stack([teams] * n), [2000 + g for g in range(n)], [g + 1 for g in range(n)], threads_per_game=tpg, **kw)
            print("keep", keep, "games", n, "threads", tpg, [many[g] == alone[g] for g in range(n)], [m[1] for m in many], [a[1] for a in alone[:n]])
        except Exception as e:
            print("keep", keep, "games", n, "threads", tpg, "ERROR", str(e)[:160])
        for c in ctxs:
            c.close()
