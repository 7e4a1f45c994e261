"""Manual GPU tool: randomized soak of the sliced search steps (oakgpu_root_steps / k_root_step) against the oracle of the crediting rule
(tests/oracle_lib.py::root_steps_reference): per-step per-root aggregates, executed turn-steps, lane streams -- all exact.
usage: root_steps_soak.py [rounds] [roots] [replicas]"""
import ctypes as C
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle_lib as O  # noqa: E402  (checker only)
from hipmem import Dev  # noqa: E402
from oak_amd import _lib  # noqa: E402
from oak_amd.engine import Context  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
roots = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(5)
bad, total = 0, 0
t0 = time.time()
for k in range(rounds):
    slice_ = int(rng.choice([16, 32, 64, 128]))
    steps = int(rng.integers(2, 5))
    b, d, p, r = O.make_random_ou_batch(roots, seed0=0x50AC5000000 + k * roots)
    adv = rng.integers(0, 60, size=roots)                       # roots at different depths of their games
    for i in np.nonzero(adv)[0]:
        out, _ = O.rollout_batch(b[i:i + 1], d[i:i + 1], r[i:i + 1], p[i:i + 1], max_steps=int(adv[i]))
        r[i] = out[0]
    lane = np.zeros((roots * reps, 8), dtype=np.uint8)
    O.LIB.oracle_fast_prng_seed_batch(O.ptr(lane), roots * reps, C.c_uint64(0xC40000000000 + k * roots * reps))
    ref_lane = lane.copy()
    cnt, s2, ex = O.root_steps_reference(b, d, r, ref_lane, reps, steps, slice_, threads=16)
    bufs = [Dev(np.ascontiguousarray(x)) for x in (b, d, r, lane)]
    report = Dev(np.zeros(roots + 2, dtype=np.uint64))
    rs = C.c_void_p()
    _lib.check(lib.oakgpu_root_steps_create(h, roots, reps, slice_, 1000, C.byref(rs)))
    ok, j = True, 0
    while True:
        _lib.check(lib.oakgpu_root_steps_launch_dev(rs, bufs[0].p, bufs[1].p, bufs[2].p, bufs[3].p, 1 if j < steps else 0, report.p))
        ctx.synchronize()
        rep = report.host()
        acc = rep[:roots]
        ok = ok and ((acc & np.uint64(0xFFFFFFFF)).astype(np.int64) == cnt[j]).all() and ((acc >> np.uint64(32)).astype(np.int64) == s2[j]).all()
        ok = ok and int(rep[roots]) == ex[j] and (rep[roots + 1] >> np.uint64(32)) == 0
        j += 1
        if (j >= steps and (rep[roots + 1] & np.uint64(0xFFFFFFFF)) == 0) or j >= cnt.shape[0]:
            break
    ok = ok and cnt[j:].sum() == 0 and (bufs[3].host() == ref_lane).all()
    lib.oakgpu_root_steps_destroy(rs)
    for x in bufs + [report]:
        x.free()
    total += int(ex.sum())
    bad += 0 if ok else 1
    print("round %d slice=%d steps=%d playouts=%d turn-steps=%d %s (%.0fs)" % (k, slice_, steps, roots * reps * steps, int(ex.sum()), "OK" if ok else "MISMATCH", time.time() - t0), flush=True)
print("root-steps soak:", "ALL EXACT" if bad == 0 else "%d rounds mismatched" % bad, "over", total, "turn-steps")
sys.exit(1 if bad else 0)
