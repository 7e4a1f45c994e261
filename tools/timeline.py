"""Manual GPU tool: wave timeline of ONE big rollout launch (profile build with -DOAKGPU_TIMELINE).
usage: timeline.py <lib.so> [n] [ppl] [rounds] -- prints when the queue ran dry and how the waves drain afterwards."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oak_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from oak_amd.engine import Context  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 20 * 65536
ppl = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
ctx.set_playouts_per_lane(ppl)
ctx.set_regroup(rounds, 32 if rounds > 1 else 0, 3)
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, prng0, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n, 8), T(n), T(n)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000 + int(os.environ.get('SEED_OFF', '0'))), n, P(battles), P(durations), P(prng0), P(rin)))
REPS = int(os.environ.get("REPS", "2"))
for rep in range(REPS):
    if rep >= 2:   # (REPS > 2: one line per earlier repetition -- the run-to-run spread of the same launch)
        wv = min(((n + 63) // 64 + ppl - 1) // ppl, 16384)
        t_ = np.zeros((wv, 5), dtype=np.uint64)
        lib.oakgpu_timeline.argtypes = [C.c_void_p, C.c_int]
        lib.oakgpu_timeline(t_.ctypes.data_as(C.c_void_p), wv)
        t_ = t_[t_[:, 0] != 0]
        e_ = (t_[:, 2].astype(np.int64) - np.int64(t_[:, 0].min())) / 100.0
        na_ = int(os.environ.get("TL_ADOPTERS", "0"))
        last_ = int(np.argmax(e_))
        print("rep %d: launch %.3f ms  bulk max exit %.0f us  adopters max exit %.0f us  last wave w%d steps %d  adopters still running 300 us before the end: %d" % (
            rep - 1, a.elapsed_time(b), e_[na_:].max(), e_[:na_].max() if na_ else 0, last_, t_[last_, 4], int((e_[:na_] > e_.max() - 300).sum()) if na_ else 0))
    prng.copy_(prng0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 1000, 0, P(rout), P(steps), P(values), None, None))
    b.record()
    torch.cuda.synchronize()
waves = min(((n + 63) // 64 + ppl - 1) // ppl, 16384)
tl = np.zeros((waves, 5), dtype=np.uint64)
lib.oakgpu_timeline.argtypes = [C.c_void_p, C.c_int]
assert lib.oakgpu_timeline(tl.ctypes.data_as(C.c_void_p), waves) == 0
tl = tl[tl[:, 0] != 0]            # (a saturated launch runs fewer waves than n / 64 / ppl: only the rows that were written)
waves = tl.shape[0]
t0 = tl[:, 0].min()
us = lambda x: (x.astype(np.int64) - np.int64(t0)) / 100.0   # 100 MHz wall clock
start, dry, end = us(tl[:, 0]), us(tl[:, 1]), us(tl[:, 2])
print("launch %.3f ms, %d waves, total steps %d (kernel counted %d)" % (a.elapsed_time(b), waves, int(steps.sum().item()), int(tl[:, 4].sum())))
print("wave start  us: min %.0f med %.0f max %.0f" % (start.min(), np.median(start), start.max()))
print("dry seen    us: min %.0f med %.0f max %.0f; live lanes at dry: mean %.1f" % (dry.min(), np.median(dry), dry.max(), tl[:, 3].mean()))
print("wave exit   us: min %.0f med %.0f p90 %.0f p99 %.0f max %.0f" % (end.min(), np.median(end), np.percentile(end, 90), np.percentile(end, 99), end.max()))
na = int(os.environ.get("TL_ADOPTERS", "0"))
if na:   # the first `na` waves are the migration's adopters (oakgpu_set_migration)
    print("adopters    us: exit min %.0f med %.0f max %.0f; steps per adopter wave: med %.0f max %.0f" % (end[:na].min(), np.median(end[:na]), end[:na].max(), np.median(tl[:na, 4]), tl[:na, 4].max()))
    print("bulk waves  us: exit med %.0f p99 %.0f max %.0f" % (np.median(end[na:]), np.percentile(end[na:], 99), end[na:].max()))
order_ = np.argsort(end)[-8:]
print("last waves   : " + "  ".join("w%d%s exit %.0f us steps %d" % (w, "(adopter)" if w < na else "", end[w], tl[w, 4]) for w in order_))
for t in range(0, min(int(end.max()) + 1000, 60000), 1000):
    print("  t=%5d us: waves still running %5d" % (t, int((end > t).sum())))
