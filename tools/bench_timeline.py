"""Manual GPU tool: the wave timeline of the bench's TIMED group launch (profile build with -DOAKGPU_TIMELINE).  bench.py runs a set-up
pass over its 20 batches and 5 warm-up batches in front of the timed 20, and the per-lane choice streams continue from pass to pass, so
the timed launch plays OTHER playouts than a fresh launch of the same battles (tools/timeline.py): this replays exactly that sequence.
usage: bench_timeline.py <lib.so> [reps]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oak_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from oak_amd.engine import Context  # noqa: E402

REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
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
off = int(os.environ.get("SEED_OFF", "0"))
for k in range(G):
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000 + off + k * n), n, P(battles[k]), P(durations[k]), P(prng0[k]), P(rin[k])))
    descs[k] = _lib.RolloutBatch(battles[k].data_ptr(), durations[k].data_ptr(), rin[k].data_ptr(), prng[k].data_ptr(), n, rout[k].data_ptr(),
                                 steps[k].data_ptr(), values[k].data_ptr(), None, None)
torch.cuda.synchronize()
na = 128
lib.oakgpu_timeline.argtypes = [C.c_void_p, C.c_int]
for rep in range(REPS):
    prng.copy_(prng0)
    for count in (20, 5):                      # bench.py's set-up pass and its --warmup 5
        _lib.check(lib.oakgpu_rollout_group_dev(h, descs, count, 1000, 0))
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(lib.oakgpu_rollout_group_dev(h, descs, G, 1000, 0))
    b.record()
    torch.cuda.synchronize()
    tl = np.zeros((4096, 5), dtype=np.uint64)
    assert lib.oakgpu_timeline(tl.ctypes.data_as(C.c_void_p), 4096) == 0
    tl = tl[tl[:, 0] != 0]
    end = (tl[:, 2].astype(np.int64) - np.int64(tl[:, 0].min())) / 100.0
    dry = (tl[:, 1].astype(np.int64) - np.int64(tl[:, 0].min())) / 100.0
    last = np.argsort(end)[-4:]
    c = np.zeros(64, dtype=np.uint32)
    _lib.check(lib.oakgpu_get_queue_counters(h, c.ctypes.data_as(C.c_void_p)))
    print("rep %d: launch %.3f ms  dry %.0f us  bulk exit p99 %.0f max %.0f  adopters max %.0f  donations %d  last waves: %s" % (
        rep, a.elapsed_time(b), np.median(dry[dry > 0]), np.percentile(end[na:], 99), end[na:].max(), end[:na].max(), int(c[40]),
        " ".join("w%d@%.0f(%d steps)" % (w, end[w], tl[w, 4]) for w in last)), flush=True)
