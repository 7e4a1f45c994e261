"""Manual GPU tool: time ONE rollout call over a large batch (default 20 x 65,536 playouts) for a sweep of queue /
regrouping settings.  usage: big_batch_sweep.py [n] -- prints one line per setting (ms, G turn-steps/s)."""
import ctypes as C
import itertools
import os
import sys

import torch

sys.path.insert(0, ".")
from oak_amd import _lib
from oak_amd.engine import Context

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20 * 65536
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, prng0, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n, 8), T(n), T(n)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng0), P(rin)))
torch.cuda.synchronize()


def run(ppl, rounds, below, shrink, reps=3):
    ctx.set_playouts_per_lane(ppl)
    ctx.set_regroup(rounds, below, shrink)
    best = 1e9
    for _ in range(reps):
        prng.copy_(prng0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 1000, 0, P(rout), P(steps), P(values), None, None))
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    tot = int(steps.sum().item())
    print("ppl %2d rounds %d below %2d shrink %d : %7.3f ms  %6.2f G turn-steps/s" % (ppl, rounds, below, shrink, best, tot / best / 1e6), flush=True)


run(5, 4, 32, 3)   # warm
grid = os.environ.get("SWEEP", "full")
if grid == "full":
    for ppl in (3, 5, 7, 10):
        run(ppl, 1, 0, 1)
    for ppl, rounds, below, shrink in itertools.product((5, 7), (4, 6, 8), (32, 48), (2, 3)):
        run(ppl, rounds, below, shrink)
