"""Manual GPU tool: per-phase cycles of k_embed_lds (wave 0 of every workgroup), -DOAKGPU_LEAF_PROFILE build."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from oak_amd import _lib, netfile

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from oak_amd.engine import Context, Network  # noqa: E402

ctx = Context(0)
ctx.ensure_ou_pools()
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
n = 65536
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, rin, rout, mid, dmid = T(n, 384), T(n, 8), T(n, 8), T(n), T(n), T(n, 384), T(n, 8)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
lib, h = ctx.lib, ctx.handle
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng), P(rin)))
_lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 20, 0, P(rout), P(steps), P(values), P(mid), P(dmid)))
netfile.write_random_net("/tmp/lp.battle.net", seed=7, hidden=256, value_hidden=256)
net = Network(ctx, path="/tmp/lp.battle.net")
buf = (C.c_ulonglong * 16)()
lib.oakgpu_leaf_profile.argtypes = [C.c_void_p, C.c_int]
_lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, P(mid), P(dmid), n, P(values), None))
torch.cuda.synchronize()
for kind, env in ((os.environ.get("OAKGPU_EMBED_KINDS", "both kinds"), None),):
    lib.oakgpu_leaf_profile(buf, 1)
    _lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, P(mid), P(dmid), n, P(values), None))
    torch.cuda.synchronize()
    lib.oakgpu_leaf_profile(buf, 0)
    print("== kinds", kind)
    names = ["prologue (image loads issued)", "-", "-", "loop top (first pass: first encode + image store + barrier)", "first layer (dense MFMA, row sums, transposition, activation)", "scatter", "next input's loads issued + second layer (split + MFMA)", "next encode (compute)"]
    wn = ["fc0 (split: 73,728 MFMA cycles per tile)", "bias + act 0", "fc1 (24,576)", "bias + act 1 (+ h1)", "value_fc2 (24,576)", "value_fc3 + sigmoid"]
    wt = sum(buf[10 + i] for i in range(6))
    for i, nm in enumerate(wn):
        print("main net wave 0: %-36s %12d cycles  %5.1f%%  (%.0f per tile)" % (nm, buf[10 + i], 100.0 * buf[10 + i] / max(wt, 1), buf[10 + i] / 512.0))
    tot = sum(buf[i] for i in range(8))
    for i, nm in enumerate(names):
        print("%-22s %12d cycles  %5.1f%%" % (nm, buf[i], 100.0 * buf[i] / tot))
