#!/usr/bin/env python3
"""Manual GPU tool: the TUTORIAL's two whole-game statistics, reproduced through the GPU path (statistical known answers of the
reference itself -- they see the engine through whole games: damage, accuracy, crits, status, Counter, Explosion, Substitute, Rest ...).

  length  `generate --budget=1024 --bandit=ucb-1.0 --policy-mode=x --eval=fp` (TUTORIAL.md:172-179) followed by `lab battle-frame-stats`
          (TUTORIAL.md:207-217): "Average battle length: 80.1695" frames per game over 242 k games.  Here: N self-play games
          (oakgpu_selfplay_games: the generator's loop, generate.cc:215-322), teams drawn uniformly from the 16 sample teams like
          TeamBuilding::Provider::get_trajectory (team-building.h:209), PokeEngine leaves, 1,024 iterations, argmax policy.
  vs      `vs --budget=4096 --bandit=ucb-1.0 --policy-mode=x --p1-eval=fp --p2-eval=mc` (TUTORIAL.md:99-104): "W D L: 186 1 31".
          Here: vs.cc's loop (vs.cc:156-345) -- both agents search every position with more than one choice, each takes ITS side's
          argmax, team pairs are played twice with the teams swapped and the agents fixed.

usage: tutorial_stats.py length [games] [batch]   |   tutorial_stats.py vs [pairs] [batch]
Output: one JSON line (also appended to gpurun_out/tutorial_stats.jsonl)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oak_amd import frames as F       # noqa: E402
from oak_amd import gamedata as G     # noqa: E402
from oak_amd import search as S       # noqa: E402
from oak_amd.engine import Context    # noqa: E402

TEAMS = json.load(open(os.path.join(ROOT, "tests", "golden", "ou_sample_teams.json")))["teams"]


def team_bytes(t):
    return np.array([[G.match_species(s[0])] + [G.match_move(m) for m in s[1:]] for s in t], dtype=np.uint8)


TB = [team_bytes(t) for t in TEAMS]
mode = sys.argv[1] if len(sys.argv) > 1 else "length"
rng = np.random.default_rng(int(os.environ.get("SEED", "20261004")))
CONC = int(os.environ.get("CONC", "16"))
ctxs = [Context(0) for _ in range(CONC)]
t0 = time.time()

if mode == "length":
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    ITER = int(os.environ.get("ITER", "1024"))                    # (sensitivity runs: the TUTORIAL's command is 1,024)
    EVAL = os.environ.get("EVAL", "poke-engine")
    lengths, results, long_games = [], [], []
    while len(lengths) < N:
        n = min(CONC, N - len(lengths))
        pick = rng.integers(0, 16, size=(n, 2))
        teams = np.stack([np.stack([TB[i], TB[j]]) for i, j in pick])
        res = F.selfplay_games(ctxs[:n], teams, rng.integers(1, 2 ** 63, size=n, dtype=np.uint64), rng.integers(1, 2 ** 31, size=n), iterations=ITER,
                               batch=batch, bandit="ucb", c=1.0, evaluator=EVAL, policy_mode=os.environ.get("POLICY", "x"))
        for (rec_bytes, frames, result), (ti, tj) in zip(res, pick):
            lengths.append(frames)
            results.append(result & 15)
            if frames >= 300 and len(long_games) < 64:
                long_games.append({"teams": [int(ti), int(tj)], "frames": int(frames), "result": int(result), "record_hex": rec_bytes.hex()})
        print("  %d games, mean length %.2f (%.0f s)" % (len(lengths), float(np.mean(lengths)), time.time() - t0), file=sys.stderr, flush=True)
    L = np.array(lengths, dtype=np.float64)
    rec = {"what": "frames per self-play game, generate --budget=1024 --bandit=ucb-1.0 --policy-mode=x --eval=fp over the 16 sample teams",
           "reference": {"average_battle_length": 80.1695282078021, "games": 242165, "where": "TUTORIAL.md:207-217"},
           "games": int(N), "batch": batch, "iterations": ITER, "eval": EVAL, "mean": float(L.mean()), "sd": float(L.std(ddof=1)), "se": float(L.std(ddof=1) / np.sqrt(N)),
           "median": float(np.median(L)), "min": int(L.min()), "max": int(L.max()),
           "percentiles_50_75_90_95_99": [float(x) for x in np.percentile(L, [50, 75, 90, 95, 99])],
           "games_of_300_frames_or_more": int((L >= 300).sum()), "games_of_1000_frames_or_more": int((L >= 1000).sum()),
           "mean_of_games_under_300_frames": float(L[L < 300].mean()),
           "results_win_lose_tie": [int(np.sum(np.array(results) == k)) for k in (1, 2, 3)], "seconds": time.time() - t0,
           "library": os.environ.get("OAKGPU_LIB", "product"), "seed": int(os.environ.get("SEED", "20261004")),
           "lengths": [int(x) for x in lengths], "results": [int(x) for x in results]}
else:
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 109          # 109 pairs = 218 games, the TUTORIAL's count
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    todo = []
    for _ in range(P):
        i, j = rng.integers(0, 16, size=2)
        todo += [(i, j), (j, i)]                                  # the pair again with the teams swapped, agents fixed (vs.cc:356-372)
    w = d = l = 0
    updates = []
    cx = ctxs[0]
    while todo:
        games = todo[:CONC]
        todo = todo[CONC:]
        n = len(games)
        teams = np.stack([np.stack([TB[i], TB[j]]) for i, j in games])
        b, dur, r = cx.battle(teams, rng.integers(1, 2 ** 63, size=n, dtype=np.uint64))
        live = np.ones(n, dtype=bool)
        ups = np.zeros(n, dtype=int)
        while live.any():
            idx = np.where(live)[0]
            c1, n1 = cx.choices(b[idx], r[idx], 0)
            c2, n2 = cx.choices(b[idx], r[idx], 1)
            pick1, pick2 = np.zeros(len(idx), dtype=int), np.zeros(len(idx), dtype=int)
            for side, (cnt, pick, ev) in enumerate(((n1, pick1, "poke-engine"), (n2, pick2, "mc"))):
                need = np.where(cnt > 1)[0]                       # (a side with one choice is not searched: vs.cc:248,263)
                if len(need) == 0:
                    continue
                outs = S.tree_search_many(ctxs[:len(need)], b[idx[need]], dur[idx[need]], r[idx[need]], rng.integers(1, 2 ** 31, size=len(need)),
                                          iterations=4096, batch=batch, c=1.0, bandit="ucb", evaluator=ev)
                for k, o in zip(need, outs):
                    pick[k] = int(np.argmax(o["p1_empirical" if side == 0 else "p2_empirical"]))   # policy mode x (policy.h:50-55)
            ch1 = c1[np.arange(len(idx)), pick1]
            ch2 = c2[np.arange(len(idx)), pick2]
            bb, dd = np.ascontiguousarray(b[idx]), np.ascontiguousarray(dur[idx])
            rr, _ = cx.update(bb, ch1, ch2, dd, want_actions=False)
            b[idx], dur[idx], r[idx] = bb, dd, rr
            ups[idx] += 1
            live[idx[(rr & 15) != 0]] = False
        for k in range(n):
            t = r[k] & 15
            w += t == 1
            l += t == 2
            d += t == 3
            updates.append(int(ups[k]))
        print("  W D L %d %d %d after %d games (%.0f s)" % (w, d, l, w + d + l, time.time() - t0), file=sys.stderr, flush=True)
    g = w + d + l
    p = (w + 0.5 * d) / g
    rec = {"what": "vs --budget=4096 --bandit=ucb-1.0 --policy-mode=x --p1-eval=fp --p2-eval=mc over the 16 sample teams (pairs played both ways)",
           "reference": {"W": 186, "D": 1, "L": 31, "score": (186 + 0.5) / 218, "where": "TUTORIAL.md:99-104"},
           "games": int(g), "batch": batch, "W": int(w), "D": int(d), "L": int(l), "score": p, "se": float(np.sqrt(p * (1 - p) / g)),
           "mean_updates": float(np.mean(updates)), "seconds": time.time() - t0}
print(json.dumps(rec))
if mode == "length" and long_games:
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(long_games, open(os.path.join(ROOT, "gpurun_out", "long_games.json"), "w"))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "tutorial_stats.jsonl"), "a").write(json.dumps(rec) + "\n")
