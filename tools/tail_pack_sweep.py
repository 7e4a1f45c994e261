"""Manual GPU tool: ONE group launch of 20 x 65,536 playouts (the driver's --steps 20 shape) for a sweep of tail-pack
settings (oakgpu_set_tail_pack: below, waves, lanes).  Prints ms per launch and G turn-steps/s per setting."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from oak_amd import _lib
from oak_amd.engine import Context

G, n = 20, 65536
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, prng0, rin, rout = T(G, n, 384), T(G, n, 8), T(G, n, 8), T(G, n, 8), T(G, n), T(G, n)
steps, values = T(G, n, dt=torch.int32), T(G, n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
descs = (_lib.RolloutBatch * G)()
for k in range(G):
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000 + int(os.environ.get('SEED_OFF', '0')) + k * n), n, P(battles[k]), P(durations[k]), P(prng0[k]), P(rin[k])))
    descs[k] = _lib.RolloutBatch(battles[k].data_ptr(), durations[k].data_ptr(), rin[k].data_ptr(), prng[k].data_ptr(), n, rout[k].data_ptr(),
                                 steps[k].data_ptr(), values[k].data_ptr(), None, None)
torch.cuda.synchronize()
ref = None


def run(below, waves, lanes, reps=3):
    global ref
    _lib.check(lib.oakgpu_set_tail_pack(h, below, waves, lanes))
    best = 1e9
    for _ in range(reps):
        prng.copy_(prng0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        a.record()
        _lib.check(lib.oakgpu_rollout_group_dev(h, descs, G, 1000, 0))
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    tot = int(steps.sum(dtype=torch.int64).item())
    sig = (tot, int(rout.sum(dtype=torch.int64).item()), int(prng.sum(dtype=torch.int64).item()))
    if ref is None:
        ref = sig
    assert sig == ref, "results changed with the schedule: %r vs %r" % (sig, ref)
    print("below %2d waves %4d lanes %2d : %7.3f ms  %6.2f G turn-steps/s" % (below, waves, lanes, best, tot / best / 1e6), flush=True)


if len(sys.argv) > 1 and sys.argv[1] == "migrate":    # long-playout migration A/B (oakgpu_set_migration)
    import numpy as np
    def counters():
        out = np.zeros(64, dtype=np.uint32)
        _lib.check(lib.oakgpu_get_queue_counters(h, out.ctypes.data_as(C.c_void_p)))
        return out
    for mode, ls, ad in ((0, 300, 0), (1, 300, 0), (1, 300, 96), (1, 300, 160), (1, 300, 192), (1, 275, 128), (1, 325, 128), (1, 350, 128), (1, 300, 256), (0, 300, 0), (1, 300, 0)):
        _lib.check(lib.oakgpu_set_migration(h, mode, ls, ad))
        print("migrate %d long_steps %3d adopters %3d" % (mode, ls, ad), end="  ")
        run(0, 0, 0, reps=4)
        c = counters()
        print("      donations %d adoptions %d bulk waves left %d errors %d" % (c[40], c[41], c[42], c[63]), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "window":     # standstill-window donation A/B (oakgpu_set_migration_window)
    import numpy as np
    def counters():
        out = np.zeros(64, dtype=np.uint32)
        _lib.check(lib.oakgpu_get_queue_counters(h, out.ctypes.data_as(C.c_void_p)))
        return out
    full = ((0, 300, 0), (64, 300, 0), (48, 300, 0), (40, 300, 0), (32, 300, 0), (24, 300, 0), (16, 300, 0), (40, 300, 192), (32, 300, 256), (24, 300, 256),
            (32, 250, 0), (40, 1000, 0), (0, 300, 0), (40, 300, 0))
    short = ((0, 300, 0), (64, 300, 0), (48, 300, 0), (32, 300, 0), (24, 300, 0), (16, 300, 0), (32, 200, 0), (48, 250, 0))
    for win, ls, ad in (short if os.environ.get("SHORT") else full):
        _lib.check(lib.oakgpu_set_migration(h, 1, ls, ad))
        _lib.check(lib.oakgpu_set_migration_window(h, win))
        print("window %3d long_steps %4d adopters %3d" % (win, ls, ad), end="  ")
        run(0, 0, 0, reps=4)
        c = counters()
        print("      donations %d adoptions %d bulk waves left %d errors %d" % (c[40], c[41], c[42], c[63]), flush=True)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "order":      # the queue-order A/B only
    for on in (0, 1, 0, 1):
        _lib.check(lib.oakgpu_set_queue_order(h, on))
        print("queue order", on, end="  ")
        run(0, 0, 0, reps=5)
    sys.exit(0)
_lib.check(lib.oakgpu_set_queue_order(h, 0))
run(0, 0, 0)
run(0, 0, 0)
for below in (4, 8, 16, 32):
    for waves in (256, 512, 1024):
        run(below, waves, 0)
for below, waves, lanes in ((8, 256, 8), (8, 512, 4), (16, 512, 8), (16, 1024, 4), (32, 1024, 8), (8, 1024, 2), (16, 2048, 2)):
    run(below, waves, lanes)
run(0, 0, 0)
