# manual GPU experiment: throughput with S batches in flight on S HIP streams
import sys, os, time, ctypes as C, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oak_amd import _lib
from oak_amd.engine import Context
os.environ['OAKGPU_ROLLOUT_BLOCK'] = sys.argv[1] if len(sys.argv) > 1 else '64'
dev = torch.device('cuda', 0)
n = 65536; u8 = torch.uint8
P = lambda t: C.c_void_p(t.data_ptr())
def make(seed0):
    ctx = Context(0); ctx.ensure_ou_pools()
    st = torch.cuda.Stream(dev); ctx.set_stream(st.cuda_stream)
    T = lambda *s, dt=u8: torch.empty(s, dtype=dt, device=dev)
    d = dict(ctx=ctx, st=st, battles=T(n, 384), durations=T(n, 8), prng=T(n, 8), rin=T(n), rout=T(n), steps=T(n, dt=torch.int32), values=T(n, dt=torch.float32))
    _lib.check(ctx.lib.oakgpu_random_ou_battles_dev(ctx.handle, C.c_uint64(seed0), n, P(d['battles']), P(d['durations']), P(d['prng']), P(d['rin'])))
    ctx.synchronize()
    return d
for S in (1, 2, 3, 4, 6, 8):
    slots = [make(0x0A4B00000000 + i * n) for i in range(S)]
    def launch(d):
        _lib.check(d['ctx'].lib.oakgpu_rollout_dev(d['ctx'].handle, P(d['battles']), P(d['durations']), P(d['rin']), P(d['prng']), n, 1000, 0, P(d['rout']), P(d['steps']), P(d['values']), None, None))
    for d in slots: launch(d)
    torch.cuda.synchronize()
    K = 24
    tot = torch.zeros((), dtype=torch.int64, device=dev)
    t0 = time.perf_counter()
    for k in range(K):
        d = slots[k % S]
        launch(d)
        with torch.cuda.stream(d['st']):
            d['acc'] = d.get('acc', torch.zeros((), dtype=torch.int64, device=dev)) + d['steps'].sum(dtype=torch.int64)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = sum(int(d['acc'].item()) for d in slots)
    print('streams %d: %d steps in %.2f ms -> %.1f M steps/s, %.3f ms/step' % (S, steps, dt * 1e3, steps / dt / 1e6, dt / K * 1e3), flush=True)
    for d in slots: d['ctx'].close()
