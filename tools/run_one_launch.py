"""Manual GPU tool: one full-size rollout launch through an alternative build of the library
(e.g. prof_build/liboakgpu_g.so with line tables for PC sampling).  usage: run_one_launch.py <lib.so> [launches]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from oak_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from oak_amd.engine import Context  # noqa: E402

reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
n = int(os.environ.get("N", 65536))
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n), T(n)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng), P(rin)))
torch.cuda.synchronize()
for _ in range(reps):
    _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 1000, 0, P(rout), P(steps), P(values), None, None))
torch.cuda.synchronize()
print("turn_steps", int(steps.sum().item()))
