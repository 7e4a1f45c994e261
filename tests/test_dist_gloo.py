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


def _config4_worker(rank, world, port, n_roots, reps, q):
    """BASELINE config 4 on two ranks: roots sharded contiguous-by-root, every root replicated `reps` times with its own
    playout streams, rolled out with root prep, reduced to one mean per root locally, ONE all-gather of the means."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oak_amd import dist as D
    lo, hi = D.root_shard(n_roots, rank, world)
    means = torch.from_numpy(_config4_means(n_roots, reps, lo, hi))
    allm = D.gather_root_means(means, n_roots)
    if rank == world - 1:
        q.put(allm.numpy())
    dist.barrier()
    dist.destroy_process_group()


def _config4_means(n_roots, reps, lo, hi):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as O
    rb, rd, rp, rr = O.make_random_ou_batch(n_roots, seed0=4000)      # the roots are global: root g is the same on any rank
    out = np.zeros(hi - lo, dtype=np.float32)
    for g in range(lo, hi):
        b, d, r = np.repeat(rb[g:g + 1], reps, 0), np.repeat(rd[g:g + 1], reps, 0), np.repeat(rr[g:g + 1], reps)
        p = np.zeros((reps, 8), dtype=np.uint8)
        for k in range(reps):                                          # replica k of root g: its own fast_prng stream
            O.LIB.oracle_fast_prng_seed(O.ptr(p[k]), C.c_uint64(0xC4000000 + g * reps + k))
        res, _ = O.rollout_batch(b, d, r, p, max_steps=150, prep=True)
        t = res & 15
        out[g - lo] = np.where(t == 1, 1.0, np.where(t == 2, 0.0, 0.5)).astype(np.float32).mean()
    return out


def test_config4_root_sharding_and_mean_gather():
    """256 roots x 4096 playouts in miniature (13 roots x 6, ragged over 2 ranks): the gathered per-root means equal the
    single-process result in global root order."""
    n_roots, reps, world = 13, 6, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config4_worker, args=(r, world, port, n_roots, reps, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert got.shape == (n_roots,) and (got == _config4_means(n_roots, reps, 0, n_roots)).all()


def _config4_groups_worker(rank, world, port, n_roots, reps, groups, q):
    """configs[3] as independent root GROUPS (oak_amd.dist.RootGroups' exchange): every rank cuts its roots into `groups`
    contiguous groups; a group's step ends with ONE all-gather of its means, padded to group_padding() floats per rank; the
    collectives are issued in a fixed (step, group) order on every rank."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oak_amd import dist as D
    lo, hi = D.root_shard(n_roots, rank, world)
    per = D.group_padding(n_roots, world, groups)
    means = _config4_means(n_roots, reps, lo, hi)
    G = max(1, min(groups, hi - lo))
    blocks = []
    for g in range(groups):                      # (every rank issues `groups` collectives, also one whose group g is empty here)
        a, b = D.shard_range(hi - lo, g, G) if g < G else (0, 0)
        padded = torch.zeros(per, dtype=torch.float32)
        padded[:b - a] = torch.from_numpy(means[a:b])
        out = torch.empty(world * per, dtype=torch.float32)
        dist.all_gather_into_tensor(out, padded)
        blocks.append(out.numpy().copy())
    if rank == 0:
        q.put(D.assemble_group_means(n_roots, world, groups, blocks))
    dist.barrier()
    dist.destroy_process_group()


def test_config4_root_groups_exchange_reassembles_the_global_root_order():
    """The grouped form of configs[3]: 13 roots ragged over 2 ranks (7 + 6), 3 groups per rank (3 + 2 + 2 and
    2 + 2 + 2 roots), one padded all-gather per group: the assembled means equal the single-process means in global root order."""
    n_roots, reps, world, groups = 13, 6, 2, 3
    from oak_amd import dist as D
    assert D.group_padding(n_roots, world, groups) == 3 and D.group_padding(256, 8, 4) == 8 and D.group_padding(256, 1, 4) == 64
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config4_groups_worker, args=(r, world, port, n_roots, reps, groups, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    assert got.shape == (n_roots,) and (got == _config4_means(n_roots, reps, 0, n_roots)).all()


# ---- configs[3] in slices (oak_amd.dist.RootSteps' exchange): per-step credited aggregates, int64, one all-gather per step -----------
def _credited(n_roots, reps, steps, slice_, lo, hi):
    """The oracle's credited aggregates of roots [lo, hi) packed the way the kernel reports them (count | sum2 << 32): lane streams are
    seeded by GLOBAL lane index, so a rank's roots give the same numbers wherever they run."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import oracle_lib as O
    rb, rd, rp, rr = O.make_random_ou_batch(n_roots, seed0=4000)
    lane = np.zeros(((hi - lo) * reps, 8), dtype=np.uint8)
    for i in range((hi - lo) * reps):
        O.LIB.oracle_fast_prng_seed(O.ptr(lane[i]), C.c_uint64(0xC40000000000 + lo * reps + i))
    cnt, s2, _ = O.root_steps_reference(rb[lo:hi], rd[lo:hi], rr[lo:hi], lane, reps, steps, slice_, max_steps=200)
    return (cnt | (s2 << 32)).astype(np.int64)          # [steps + tail, hi - lo]


def _root_steps_worker(rank, world, port, n_roots, reps, steps, slice_, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oak_amd import dist as D
    lo, hi = D.root_shard(n_roots, rank, world)
    per = -(-n_roots // world)
    mine = _credited(n_roots, reps, steps, slice_, lo, hi)
    got = []
    for k in range(mine.shape[0]):               # one collective per step, the same count on every rank (padded)
        send = torch.zeros(per, dtype=torch.int64)
        send[:hi - lo] = torch.from_numpy(mine[k])
        recv = torch.empty(world * per, dtype=torch.int64)
        dist.all_gather_into_tensor(recv, send)
        got.append(D.assemble_rank_blocks(n_roots, world, per, recv.numpy()))
    if rank == world - 1:
        q.put(np.stack(got))
    dist.barrier()
    dist.destroy_process_group()


def test_config4_sliced_steps_gather_the_credited_aggregates_in_global_root_order():
    """13 roots ragged over 2 ranks (7 + 6), 3 search steps in slices of 16 turn-steps: every step's gathered (count, 2 x value sum) per
    root -- late credits and drain steps included -- equals the single-process oracle's, and every playout is credited exactly once."""
    n_roots, reps, steps, slice_, world = 13, 6, 3, 16, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_root_steps_worker, args=(r, world, port, n_roots, reps, steps, slice_, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=180)
    for p in procs:
        p.join(timeout=180)
        assert p.exitcode == 0
    want = _credited(n_roots, reps, steps, slice_, 0, n_roots)
    assert got.shape == want.shape and (got == want).all()
    assert int((got & 0xFFFFFFFF).sum()) == n_roots * reps * steps
    from oak_amd import dist as D
    m = D.credited_means(got[0] & 0xFFFFFFFF, got[0] >> 32)
    assert m.shape == (n_roots,) and ((m >= 0) & (m <= 1)).all()
