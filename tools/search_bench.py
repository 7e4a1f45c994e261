"""Manual GPU tool: throughput of the batched-leaf tree search (oakgpu_search) on a random OU position."""
import json
import os
import sys

import numpy as np  # noqa: F401

sys.path.insert(0, ".")
from oak_amd import netfile
from oak_amd.engine import Context, Network
from oak_amd.search import tree_search

import ctypes as C

import torch

from oak_amd import _lib

ctx = Context(0)
ctx.ensure_ou_pools()
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
tb, td, tp, tr = (torch.empty(s_, dtype=torch.uint8, device=dev) for s_ in ((1, 384), (1, 8), (1, 8), (1,)))
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(ctx.lib.oakgpu_random_ou_battles_dev(ctx.handle, C.c_uint64(0x0A4B00000000), 1, P(tb), P(td), P(tp), P(tr)))
torch.cuda.synchronize()
b, d, r = tb.cpu().numpy(), td.cpu().numpy(), tr.cpu().numpy()
path = "/tmp/search_bench.battle.net"
netfile.write_random_net(path, seed=7, hidden=256, value_hidden=256)
net = Network(ctx, path=path)
rows = []
for ev, name in (("mc", "monte-carlo"), (net, "network-256")):
    for batch in (1024, 4096, 16384):
        it = 1 << 18 if ev == "mc" else 1 << 19
        tree_search(ctx, b[0], d[0], int(r[0]), iterations=batch * 2, batch=batch, evaluator=ev)      # warm-up
        out = tree_search(ctx, b[0], d[0], int(r[0]), iterations=it, batch=batch, evaluator=ev, bandit="ucb")
        rows.append({"eval": name, "batch": batch, "iterations": out["iterations"], "ms": out["duration_ms"],
                     "iterations_per_s": out["iterations"] / out["duration_ms"] * 1e3, "nodes": out["nodes"],
                     "mean_depth": out["mean_depth"], "nash_value": out["nash_value"]})
        print(json.dumps(rows[-1]), flush=True)
net.close()
