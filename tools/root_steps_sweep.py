#!/usr/bin/env python3
"""configs[3] in slices: ms per search step and turn-steps/s of oakgpu_root_steps by roots in flight and slice length (raw C ABI,
device buffers through tests/hipmem.py).  usage: tools/root_steps_sweep.py [roots,roots,...] [slice,slice,...] [steps] [replicas per root]
(replicas = 1 with 2^20 roots: the headline's heterogeneous workload -- every playout another team pair -- through this kernel)"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from hipmem import Dev  # noqa: E402
from oak_amd import _lib  # noqa: E402
from oak_amd.engine import Context  # noqa: E402

roots_list = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32,256").split(",")]
slices = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "64,128").split(",")]
K = int(sys.argv[3]) if len(sys.argv) > 3 else 10
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4096
MAX_STEPS = int(os.environ.get("SWEEP_MAX_STEPS", "1000"))       # (a cap of 250 leaves no tail: the steady rate of a launch without slices)
ctx = Context(0)
ctx.ensure_ou_pools()
lib, h = ctx.lib, ctx.handle
out = []
for roots in roots_list:
    n = roots * reps
    rb, rd, rp, rr = (Dev(np.zeros(s_, dtype=np.uint8)) for s_ in ((roots, 384), (roots, 8), (roots, 8), (roots,)))
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), roots, rb.p, rd.p, rp.p, rr.p))
    for slice_ in slices:
        tb, tdur, tr, lane = (Dev(np.zeros(s_, dtype=np.uint8)) for s_ in ((n, 384), (n, 8), (n,), (n, 8)))
        _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0xC40000000000), n, tb.p, tdur.p, lane.p, tr.p))
        ctx.synchronize()
        for x in (tb, tdur, tr):
            x.free()
        rs = C.c_void_p()
        _lib.check(lib.oakgpu_root_steps_create(h, roots, reps, slice_, MAX_STEPS, C.byref(rs)))
        report = Dev(np.zeros(roots + 2, dtype=np.uint64))
        warm = 2 + (256 // slice_ if slice_ else 0)
        for _ in range(warm):
            _lib.check(lib.oakgpu_root_steps_launch_dev(rs, rb.p, rd.p, rr.p, lane.p, 1, report.p))
        ctx.synchronize()
        steps, per = 0, []
        t0 = time.perf_counter()
        for _ in range(K):
            t1 = time.perf_counter()
            _lib.check(lib.oakgpu_root_steps_launch_dev(rs, rb.p, rd.p, rr.p, lane.p, 1, report.p))
            ctx.synchronize()
            per.append(time.perf_counter() - t1)
            rep = report.host()
            steps += int(rep[roots])
        dt = time.perf_counter() - t0
        carried = int(rep[roots + 1] & np.uint64(0xFFFFFFFF))
        rec = {"roots": roots, "slice": slice_, "ms_per_step": dt / K * 1e3, "min_ms": min(per) * 1e3, "turn_steps_per_s": steps / dt,
               "turn_steps_per_step": steps / K, "carried": carried, "credited_last": int((rep[:roots] & np.uint64(0xFFFFFFFF)).sum()), "err": int(rep[roots + 1] >> np.uint64(32))}
        print(json.dumps(rec), flush=True)
        out.append(rec)
        lib.oakgpu_root_steps_destroy(rs)
        report.free()
        lane.free()
    for x in (rb, rd, rp, rr):
        x.free()
ctx.close()
