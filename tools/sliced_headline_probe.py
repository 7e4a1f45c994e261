#!/usr/bin/env python3
"""Manual GPU probe: what would the headline's group launch (20 x 65,536 playouts, every playout another team pair) cost as a sequence of
SLICED launches -- one fresh launch of 1.31 M playouts, then drain launches until nobody is left?  (oakgpu_root_steps with one replica per
root; its outputs are not the rollout API's, only the time is of interest.)  usage: tools/sliced_headline_probe.py [slice,slice,...]"""
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

slices = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16,32,64,128").split(",")]
roots = 20 * 65536
ctx = Context(0)
ctx.ensure_ou_pools()
lib, h = ctx.lib, ctx.handle
rb, rd, rp, rr = (Dev(np.zeros(s_, dtype=np.uint8)) for s_ in ((roots, 384), (roots, 8), (roots, 8), (roots,)))
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), roots, rb.p, rd.p, rp.p, rr.p))
lane0 = rp.host()
for slice_ in slices:
    rs = C.c_void_p()
    _lib.check(lib.oakgpu_root_steps_create(h, roots, 1, slice_, 1000, C.byref(rs)))
    report = Dev(np.zeros(roots + 2, dtype=np.uint64))
    lane = Dev(lane0)
    best = None
    for rep_ in range(3):
        lane.put(lane0)
        ctx.synchronize()
        t0 = time.perf_counter()
        launches = 0
        _lib.check(lib.oakgpu_root_steps_launch_dev(rs, rb.p, rd.p, rr.p, lane.p, 1, report.p))
        for _ in range((1000 + slice_ - 1) // slice_):      # enough drain launches for a playout that runs into the cap, enqueued blind
            _lib.check(lib.oakgpu_root_steps_launch_dev(rs, rb.p, rd.p, rr.p, lane.p, 0, report.p))
            launches += 1
        ctx.synchronize()
        dt = time.perf_counter() - t0
        carried = int(report.host()[roots + 1] & np.uint64(0xFFFFFFFF))
        best = dt if best is None or dt < best else best
    print(json.dumps({"slice": slice_, "launches": launches + 1, "total_ms": best * 1e3, "carried_after": carried,
                      "G_turn_steps_per_s_if_129.7M": 129.7e6 / best / 1e9}), flush=True)
    lib.oakgpu_root_steps_destroy(rs)
    report.free()
    lane.free()
ctx.close()
