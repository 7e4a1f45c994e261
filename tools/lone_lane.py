"""Manual GPU tool: per-step latency of a LONE playout (the critical path of a group launch's tail): lane `SEED_OFF` of the
config-2 batch (46 is a frozen-vs-Struggle stalemate that runs to the 1000-step cap), one wave, nothing else on the GPU."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from oak_amd import _lib
from oak_amd.engine import Context

ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
n = int(os.environ.get("N", 1))
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, prng0, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n, 8), T(n), T(n)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000 + int(os.environ.get("SEED_OFF", 46))), n, P(battles), P(durations), P(prng0), P(rin)))
best = 1e9
for _ in range(5):
    prng.copy_(prng0)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 1000, 0, P(rout), P(steps), P(values), None, None))
    b.record()
    torch.cuda.synchronize()
    best = min(best, a.elapsed_time(b))
s = int(steps.sum().item())
print("n %d steps %d  %.3f ms  %.2f us per step of the longest playout" % (n, s, best, best * 1e3 / int(steps.max().item())))
