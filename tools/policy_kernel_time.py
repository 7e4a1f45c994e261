"""Manual GPU tool: wall time of value_inference vs value_policy_inference on device-resident inputs (65,536 mid-game states,
768-256-256-256-1 net with 64-wide policy heads): the difference is k_policy.  usage: python tools/policy_kernel_time.py [lib.so]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, ".")
from oak_amd import _lib, netfile

if len(sys.argv) > 1:
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
c1, c2, n1, n2 = T(n, 9), T(n, 9), T(n), T(n)
l1, l2 = T(n, 9, dt=torch.float32), T(n, 9, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
lib, h = ctx.lib, ctx.handle
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng), P(rin)))
_lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 20, 0, P(rout), P(steps), P(values), P(mid), P(dmid)))
_lib.check(lib.oakgpu_choices_dev(h, P(mid), P(rout), 0, P(c1), P(n1), n))
_lib.check(lib.oakgpu_choices_dev(h, P(mid), P(rout), 1, P(c2), P(n2), n))
netfile.write_random_net("/tmp/pk.battle.net", seed=7, hidden=256, value_hidden=256)
net = Network(ctx, path="/tmp/pk.battle.net")


def timed(f, reps=20):
    for _ in range(5):
        f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


tv = timed(lambda: _lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, P(mid), P(dmid), n, P(values), None)))
tp = timed(lambda: _lib.check(lib.oakgpu_leaf_eval_policy_dev(h, net.handle, P(mid), P(dmid), n, P(c1), P(n1), P(c2), P(n2), P(values), P(l1), P(l2))))
print("value_inference %.1f us   value_policy_inference %.1f us   (policy heads: %.1f us; %.1f M policy leaf-evals/s)" % (tv, tp, tp - tv, n / tp))
