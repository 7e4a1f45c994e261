# manual GPU experiment: A/B of the two rollout engines (time + parity of outputs between them)
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oak_amd import _lib
from oak_amd.engine import Context
def run(engine, block, caps, n=65536, keep=True):
    os.environ['OAKGPU_ROLLOUT_BLOCK'] = str(block); os.environ['OAKGPU_ROLLOUT_ENGINE'] = str(engine)
    ctx = Context(0); lib, h = ctx.lib, ctx.handle
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream(dev); ctx.set_stream(stream.cuda_stream); ctx.ensure_ou_pools()
    u8 = torch.uint8
    T = lambda *s, dt=u8: torch.empty(s, dtype=dt, device=dev)
    battles, durations, prng, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n), T(n)
    steps, values, bout = T(n, dt=torch.int32), T(n, dt=torch.float32), T(n, 384)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng), P(rin)))
    torch.cuda.synchronize()
    prng0 = prng.clone()
    res = {}
    for cap in caps:
        ts = []
        for it in range(4):
            prng.copy_(prng0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, cap, 0, P(rout), P(steps), P(values), P(bout) if keep else None, None))
            b.record(stream); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        tot = int(steps.sum().item())
        if keep: res[cap] = (bout.cpu().numpy().copy(), steps.cpu().numpy().copy(), rout.cpu().numpy().copy())
        print('engine %d block %3d n %7d cap %4d: %.3f ms  steps %d  -> %.1f M steps/s' % (engine, block, n, cap, min(ts[1:]), tot, tot / min(ts[1:]) / 1e3), flush=True)
    ctx.close()
    return res
caps = [25, 300, 1000]
r1 = run(1, 256, caps)
r2 = run(2, 128, caps)
for cap in caps:
    print('cap', cap, 'engines agree:', (r1[cap][0] == r2[cap][0]).all(), (r1[cap][1] == r2[cap][1]).all(), (r1[cap][2] == r2[cap][2]).all())
run(2, 64, caps, keep=False)
run(2, 128, [25, 1000], n=131072, keep=False)
run(2, 128, [25, 1000], n=262144, keep=False)
