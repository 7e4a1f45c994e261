# manual GPU experiment (not a pytest file): kernel time vs step cap / block size
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oak_amd import _lib
from oak_amd.engine import Context
def run(block, caps):
    os.environ['OAKGPU_ROLLOUT_BLOCK'] = str(block)
    ctx = Context(0); lib, h = ctx.lib, ctx.handle
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream(dev); ctx.set_stream(stream.cuda_stream); ctx.ensure_ou_pools()
    n = 65536; u8 = torch.uint8
    T = lambda *s, dt=u8: torch.empty(s, dtype=dt, device=dev)
    battles, durations, prng, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n), T(n)
    steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
    P = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000), n, P(battles), P(durations), P(prng), P(rin)))
    torch.cuda.synchronize()
    for cap in caps:
        ts = []
        for it in range(4):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, cap, 0, P(rout), P(steps), P(values), None, None))
            b.record(stream); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        tot = int(steps.sum().item())
        print('block %d cap %4d: %.3f ms  steps %d  -> %.1f M steps/s' % (block, cap, min(ts[1:]), tot, tot / min(ts[1:]) / 1e3), flush=True)
    ctx.close()
caps = [25, 50, 100, 200, 300, 1000]
run(256, caps)
run(64, caps)
