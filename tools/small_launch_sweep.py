"""Manual GPU tool: ONE rollout launch that does NOT fill the device -- e.g. a rank's share of configs[3] at 8 GPUs: 32 roots x 4,096
playouts with root prep -- for a sweep of schedules (regrouping rounds / forced queue order / forced migration).
usage: small_launch_sweep.py [roots] [replicas per root] [prep]"""
import ctypes as C
import sys

import torch

sys.path.insert(0, ".")
from oak_amd import _lib
from oak_amd.engine import Context

roots = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
prep = int(sys.argv[3]) if len(sys.argv) > 3 else 1
n = roots * reps
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
rb, rd, rp, rr = T(roots, 384), T(roots, 8), T(roots, 8), T(roots)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), roots, P(rb), P(rd), P(rp), P(rr)))
torch.cuda.synchronize()
battles, durations, rin = rb.repeat_interleave(reps, 0).contiguous(), rd.repeat_interleave(reps, 0).contiguous(), rr.repeat_interleave(reps, 0).contiguous()
tb, td, prng0, tr = T(n, 384), T(n, 8), T(n, 8), T(n)
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0xC40000000000), n, P(tb), P(td), P(prng0), P(tr)))
torch.cuda.synchronize()
prng, rout, steps, values = T(n, 8), T(n), T(n, dt=torch.int32), T(n, dt=torch.float32)
ref = None


def run(label, reps_=4):
    global ref
    best = 1e9
    for _ in range(reps_):
        prng.copy_(prng0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 1000, prep, P(rout), P(steps), P(values), None, None))
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    tot = int(steps.sum(dtype=torch.int64).item())
    sig = (tot, int(rout.sum(dtype=torch.int64).item()), int(prng.sum(dtype=torch.int64).item()))
    if ref is None:
        ref = sig
    assert sig == ref, "results changed with the schedule"
    print("%-64s %7.3f ms  %6.2f G turn-steps/s" % (label, best, tot / best / 1e6), flush=True)


run("default")
run("default")
if len(sys.argv) > 4 and sys.argv[4] == "spread":
    for ppl in (2, 1):
        ctx.set_playouts_per_lane(ppl)
        for lanes in (0, 48, 32, 24, 16, 12, 8, 4):
            _lib.check(lib.oakgpu_set_spread(h, lanes))
            run("playouts per lane %d, spread: %s lanes per wave" % (ppl, lanes or "all 64"))
    sys.exit(0)
_lib.check(lib.oakgpu_set_queue_order(h, 0))
_lib.check(lib.oakgpu_set_regroup(h, 4, 32, 3))
run("regrouping rounds 4 / 32 / 3, queue order 0")
_lib.check(lib.oakgpu_set_regroup(h, 1, 0, 1))
run("single dispatch")
for order in (0, 1):
    _lib.check(lib.oakgpu_set_queue_order(h, order))
    for ad, ls in ((0, 300), (32, 300), (64, 300), (128, 300), (64, 200), (64, 400)):
        _lib.check(lib.oakgpu_set_migration(h, 2 if ad else 0, ls, ad))
        run("single dispatch, queue order %d, migration %s" % (order, "adopters %d long %d" % (ad, ls) if ad else "off"))
_lib.check(lib.oakgpu_set_queue_order(h, 1))
_lib.check(lib.oakgpu_set_migration(h, 0, 300, 0))
_lib.check(lib.oakgpu_set_regroup(h, 4, 32, 3))
run("regrouping rounds 4 / 32 / 3 + queue order 1")
