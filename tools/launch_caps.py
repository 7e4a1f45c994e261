# manual GPU experiment for rocprofv3 --pmc: one launch per cap so counters can be differenced
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oak_amd import _lib
from oak_amd.engine import Context
ctx = Context(0); lib, h = ctx.lib, ctx.handle
dev = torch.device('cuda', 0)
stream = torch.cuda.current_stream(dev); ctx.set_stream(stream.cuda_stream); ctx.ensure_ou_pools()
n = 65536; u8 = torch.uint8
T = lambda *s, dt=u8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n), T(n)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng), P(rin)))
torch.cuda.synchronize()
prng0 = prng.clone()
for cap in [int(x) for x in sys.argv[1:]]:
    prng.copy_(prng0)
    _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, cap, 0, P(rout), P(steps), P(values), None, None))
    torch.cuda.synchronize()
    print('cap', cap, 'steps', int(steps.sum().item()))
