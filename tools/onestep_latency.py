import ctypes as C, sys, os
import torch
sys.path.insert(0, ".")
from oak_amd import _lib
from oak_amd.engine import Context
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
n = 65536
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, rin = T(n, 384), T(n, 8), T(n, 8), T(n)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng), P(rin)))
b0, d0, p0, r0 = battles.clone(), durations.clone(), prng.clone(), rin.clone()
for ppl in (1,):
    _lib.check(lib.oakgpu_set_playouts_per_lane(h, ppl))
    for ms in (0, 1, 2, 4, 8):
        best = 1e9
        for rep in range(4):
            battles.copy_(b0); durations.copy_(d0); prng.copy_(p0); rin.copy_(r0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            for _ in range(10):
                _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, ms, 0, P(rin), P(steps), P(values), P(battles), P(durations)))
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / 10)
        print("ppl %d max_steps %d: %.1f us per launch" % (ppl, ms, best * 1e3), flush=True)
