"""Manual GPU tool: are concurrent searches (oakgpu_search_many) reproducible, and equal to the searches run alone?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from oak_amd.engine import Context
from oak_amd.search import tree_search, tree_search_many

n = 6
b, d, p, r = O.make_random_ou_batch(n, seed0=0x51DE5)
ctxs = [Context(0) for _ in range(n)]
main = Context(0)
seeds = [1000 + 17 * i for i in range(n)]
kw = dict(evaluator="mc", bandit="ucb", c=2.0, iterations=1 << 13, batch=1024)
runs = [tree_search_many(ctxs, b, d, r, seeds, threads_per_search=int(os.environ.get("TPS", "2")), **kw) for _ in range(3)]
for k in (1, 2):
    print("many run %d vs run 0:" % k, [bool((runs[k][i]["visit_matrix"] == runs[0][i]["visit_matrix"]).all()) for i in range(n)])
alone = []
for rep in range(2):
    alone.append([tree_search(main, b[i], d[i], int(r[i]), seed=seeds[i], **kw) for i in range(n)])
print("alone run 1 vs run 0:", [bool((alone[1][i]["visit_matrix"] == alone[0][i]["visit_matrix"]).all()) for i in range(n)])
print("many vs alone:", [bool((runs[0][i]["visit_matrix"] == alone[0][i]["visit_matrix"]).all()) for i in range(n)])
fresh = [tree_search(ctxs[i], b[i], d[i], int(r[i]), seed=seeds[i], **kw) for i in range(n)]
print("alone on the many-contexts vs alone on main:", [bool((fresh[i]["visit_matrix"] == alone[0][i]["visit_matrix"]).all()) for i in range(n)])
