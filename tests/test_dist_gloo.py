"""CPU, world_size 2 over gloo: the N>1 path (lane sharding + the single all-gather of leaf values).
The per-rank 'rollout' here is the CPU oracle standing in for the GPU kernel; what is under test is
oak_amd.dist (sharding, seeds, gather order), which bench.py uses unchanged with backend nccl."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    from oak_amd import dist as D
    lo, hi = D.shard_range(n_total, rank, world)
    b, d, p, r = O.make_random_ou_batch(hi - lo, seed0=D.lane_seed0(1000, n_total, rank, world))
    out, steps = O.rollout_batch(b, d, r, p, max_steps=200)
    t = out & 15
    vals = torch.from_numpy(np.where(t == 1, 1.0, np.where(t == 2, 0.0, 0.5)).astype(np.float32))
    allv = D.gather_values(vals, n_total)
    if rank == 0:
        q.put(allv.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _single(n_total):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    b, d, p, r = O.make_random_ou_batch(n_total, seed0=1000)
    out, _ = O.rollout_batch(b, d, r, p, max_steps=200)
    t = out & 15
    return np.where(t == 1, 1.0, np.where(t == 2, 0.0, 0.5)).astype(np.float32)


def _run(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    return got


def test_two_ranks_equal_single_process_even():
    assert (_run(256) == _single(256)).all()


def test_two_ranks_equal_single_process_ragged():
    assert (_run(257) == _single(257)).all()


def _round_worker(rank, world, port, S, n, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oak_amd import dist as D
    # batch k of rank r holds the values of global lanes [r * n, (r + 1) * n) of batch k: value = 1000 k + global lane
    stage = torch.stack([1000.0 * k + torch.arange(rank * n, (rank + 1) * n, dtype=torch.float32) for k in range(S)])
    out, work = D.gather_round(stage, async_op=True)
    work.wait()
    if rank == 1:
        q.put(D.round_to_global(out).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_round_gather_puts_every_batch_in_global_lane_order():
    """bench.py's exchange: ONE all-gather per round of S batches in flight ([S, n] per rank -> [world, S, n])."""
    S, n, world = 5, 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_round_worker, args=(r, world, port, S, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    expect = np.stack([1000.0 * k + np.arange(world * n, dtype=np.float32) for k in range(S)])
    assert got.shape == (S, world * n) and (got == expect).all()


def test_shard_range_partition():
    from oak_amd.dist import shard_range
    for n in (0, 1, 7, 64, 65536, 65537):
        for w in (1, 2, 3, 8):
            edges = [shard_range(n, r, w) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
