"""Manual GPU tool: where one tree search spends its time (OAKGPU_SEARCH_TIMING=1): BASELINE configs[4]'s single-root shape -- 2^18
iterations in batches of 16,384 descents, 768-256-256-256-1 network leaves, joint UCB.  usage: search_timing.py [iterations] [batch] [repeats]"""
import json
import os
import sys
import time

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
os.environ["OAKGPU_SEARCH_TIMING"] = "1"
import oracle_lib as O  # noqa: E402
from oak_amd.engine import Context, Network  # noqa: E402
from oak_amd.search import tree_search  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = Context(0)
b, d, p, r = O.make_random_ou_batch(1, seed0=0x0A4B00000000)
net = Network(ctx, path=os.path.join("tests", "golden", "net_256.battle.net"))
tree_search(ctx, b[0], d[0], int(r[0]), iterations=2 * batch, batch=batch, evaluator=net)
for k in range(reps):
    if len(sys.argv) > 4:                         # thread counts to compare, e.g. 8,16
        os.environ["OAKGPU_SEARCH_THREADS"] = sys.argv[4].split(",")[k % len(sys.argv[4].split(","))]
    t0 = time.perf_counter()
    out = tree_search(ctx, b[0], d[0], int(r[0]), iterations=it, batch=batch, evaluator=net, seed=k)
    dt = time.perf_counter() - t0
    print(json.dumps({"iterations": out["iterations"], "ms": out["duration_ms"], "wall_ms": dt * 1e3, "iterations_per_s": out["iterations"] / out["duration_ms"] * 1e3,
                      "nodes": out["nodes"], "mean_depth": out["mean_depth"]}), flush=True)
net.close()
