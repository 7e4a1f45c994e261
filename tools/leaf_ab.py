"""Manual GPU tool: time value_inference (65,536 mid-game leaves, 768-256-256-256-1) with alternative builds of the library.
usage: leaf_ab.py lib1.so [lib2.so ...]   (each library is timed in its own child process)"""
import ctypes as C
import os
import subprocess
import sys


def run(path):
    import torch
    sys.path.insert(0, ".")
    from oak_amd import _lib, netfile
    _lib.LIB_PATH = os.path.abspath(path)
    from oak_amd.engine import Context, Network
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
    netfile.write_random_net("/tmp/ab.battle.net", seed=7, hidden=256, value_hidden=256)
    net = Network(ctx, path="/tmp/ab.battle.net")
    for _ in range(5):
        _lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, P(mid), P(dmid), n, P(values), None))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    K = 40
    for _ in range(K):
        _lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, P(mid), P(dmid), n, P(values), None))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    lib.oakgpu_set_kernel_timing(h, 1)
    acc = [0.0, 0.0, 0.0]
    buf = (C.c_float * 3)()
    for _ in range(5):
        _lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, P(mid), P(dmid), n, P(values), None))
        torch.cuda.synchronize()
        lib.oakgpu_get_leaf_kernel_ms(h, buf)
        for i in range(3):
            acc[i] += buf[i] / 5 * 1e3
    print("%-40s %.1f us per call = %.1f M leaf-evals/s; kernels (party, actives, main net) %.0f %.0f %.0f us; checksum %.6f"
          % (os.path.basename(path), ms * 1e3, n / ms / 1e3, acc[0], acc[1], acc[2], float(values.double().sum())), flush=True)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        run(sys.argv[2])
    else:
        for p in sys.argv[1:]:
            subprocess.run([sys.executable, __file__, "--child", p], check=False)
