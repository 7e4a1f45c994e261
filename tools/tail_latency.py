"""Manual GPU tool: per-step latency of a capped (stalemate) playout when W waves each hold ONE live lane -- the shape of a
group launch's tail.  usage: tail_latency.py [W ...]  (lane 0 of every wave = the stalemate of tools/lone_lane.py, the
other 63 lanes already terminal; one playout per lane)."""
import ctypes as C
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
_lib.check(lib.oakgpu_set_playouts_per_lane(h, 1))
P = lambda t: C.c_void_p(t.data_ptr())
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
b1, d1, p1, r1 = T(1, 384), T(1, 8), T(1, 8), T(1)
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000 + 46), 1, P(b1), P(d1), P(p1), P(r1)))
for W in [int(x) for x in sys.argv[1:]] or [1, 256, 1024, 4096]:
    n = 64 * W
    battles, durations, prng, rin, rout = b1.repeat(n, 1), d1.repeat(n, 1), p1.repeat(n, 1), torch.ones(n, dtype=torch.uint8, device=dev), T(n)
    rin[::64] = r1[0]
    steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
    best = 1e9
    for _ in range(3):
        pr = prng.clone()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(pr), n, 1000, 0, P(rout), P(steps), P(values), None, None))
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    st = steps.to(torch.int64)
    print("waves %5d  live lanes %d  total steps %d  max %d  %.3f ms  = %.2f us per step" % (W, int((st > 0).sum()), int(st.sum()), int(st.max()), best, best * 1e3 / max(int(st.max()), 1)), flush=True)
