#!/usr/bin/env python3
"""configs[3] as independent root groups (oak_amd.dist.RootGroups): ms per search step and turn-steps/s for the full job (256 roots x
4,096 playouts on one GPU) and for a rank's share at 8 GPUs (32 roots), by number of groups; plus how many roots a GPU needs in
flight before the step stops being bound by its longest playout.  -> gpurun_out/r04/config4_pipeline.json
usage: tools/config4_pipeline.py [steps]"""
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")       # every group's stream on a hardware queue of its own (as bench.py)
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oak_amd import _lib  # noqa: E402
from oak_amd import dist as oakdist  # noqa: E402
from oak_amd.engine import Context  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
SEED0, reps = 0x0A4B00000000, 4096
ctx = Context(0)
ctx.ensure_ou_pools()
lib, h = ctx.lib, ctx.handle
u8 = torch.uint8


def P(t):
    return C.c_void_p(t.data_ptr())


SPREAD = int(os.environ.get("SPREAD", "-1"))          # oakgpu_set_spread: -1 automatic, 0 off
PPL = int(os.environ.get("PPL", "2"))
REGROUP = os.environ.get("REGROUP", "")                # "rounds,below,shrink"
MIGRATE = os.environ.get("MIGRATE", "")                # "mode,long_steps,adopters"


def make_context():
    c = Context(0)
    c.ensure_ou_pools()
    _lib.check(c.lib.oakgpu_set_spread(c.handle, SPREAD))
    c.set_playouts_per_lane(PPL)
    if REGROUP:
        c.set_regroup(*[int(x) for x in REGROUP.split(",")])
    if MIGRATE:
        c.set_migration(*[int(x) for x in MIGRATE.split(",")])
    return c


def setup(n_roots):
    rb, rd, rp, rr = (torch.empty(s_, dtype=u8, device=dev) for s_ in ((n_roots, 384), (n_roots, 8), (n_roots, 8), (n_roots,)))
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(SEED0), n_roots, P(rb), P(rd), P(rp), P(rr)))
    ctx.synchronize()
    n = n_roots * reps
    battles, durations, rin = rb.repeat_interleave(reps, 0).contiguous(), rd.repeat_interleave(reps, 0).contiguous(), rr.repeat_interleave(reps, 0).contiguous()
    prng = torch.empty((n, 8), dtype=u8, device=dev)
    tb, tdur, tr = torch.empty((n, 384), dtype=u8, device=dev), torch.empty((n, 8), dtype=u8, device=dev), torch.empty((n,), dtype=u8, device=dev)
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0xC40000000000), n, P(tb), P(tdur), P(prng), P(tr)))
    ctx.synchronize()
    return battles, durations, rin, prng


rows = []
for n_roots in [int(x) for x in os.environ.get("ROOTS", "256,128,64,32").split(",")]:
    battles, durations, rin, prng0 = setup(n_roots)
    for G in [int(x) for x in os.environ.get("GROUPS", "1,2,4,8,16").split(",")]:
        if G > n_roots // 4:
            continue
        prng = prng0.clone()
        rg = oakdist.RootGroups(make_context, dev, battles, durations, rin, prng, n_roots, reps, G)
        rg.run(2)                                  # warm-up (contexts allocate their tables and scratch)
        rg.total.zero_()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        rg.run(K)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        steps = int(rg.total.sum().item())
        rows.append({"roots": n_roots, "groups": G, "ms_per_step": dt / K * 1e3, "g_turn_steps_per_s": steps / dt / 1e9, "spread": SPREAD, "ppl": PPL,
                     "regroup": REGROUP, "migrate": MIGRATE})
        print(rows[-1], flush=True)
        rg.close()
    del battles, durations, rin, prng0
os.makedirs("gpurun_out/r04", exist_ok=True)
json.dump({"steps": K, "playouts_per_root": reps, "rows": rows}, open(os.environ.get("OUT", "gpurun_out/r04/config4_pipeline.json"), "w"), indent=1)
