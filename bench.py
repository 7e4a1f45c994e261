#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path on MI355X.

Headline workload (BASELINE.json configs[1]): batch = 65,536 random OU team pairs per GPU, pure random-policy
rollout to terminal (cap 1000 turn-steps, mirrors search/mcts.h:606-614).  One *step* = one pass of the rollout
kernel over one whole batch; inputs (battles, durations, per-lane fast_prng state) are generated ON DEVICE before the
timed region and stay resident in HBM; the per-lane choice-RNG stream continues from pass to pass, so every step
plays different playouts.  Steps are independent batches (as in root-parallel MCTS) and are submitted in GROUPS of up
to `--group` batches per launch (oakgpu_rollout_group_dev): one playout queue, one tail per group.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, lanes sharded by rank (weak scaling: 65,536 playouts per GPU and step, disjoint seeds); the
path's single exchange is an RCCL all-gather of the fp32 leaf values back to every root -- ONE collective per group
launch (G x 256 KiB per rank), on its own stream.

Prints ONE JSON line (rank 0).  The line is the configs[1] record; with the default `--workload all` it also carries
the rest of BASELINE.json's metric as sub-records with their own timed regions: `leaf` (leaf-evals/s of the
768-256-256-256-1 network, fp32-MFMA roofline, per-kernel microseconds) and `config3` (configs[2]: turn-step + leaf
evaluation every turn).  `roofline` of the headline record prices the rollout kernel against the HBM roof with the
ALGORITHMIC bytes of SURVEY 8(d): 802 B per turn-step (401 read + 401 written of per-lane state); every record has a
`cpu_baseline` at N = 1: the CPU oracle (a restatement, not the Oak binary) on all host cores.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", os.environ.get("BENCH_HWQ", "32"))   # let every in-flight batch have its own hardware queue
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_STEP = 802          # 384 battle + 8 durations + 8 rng + 1 result, read + written
HBM_PEAK_GBPS = 8000.0             # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
SEED0 = 0x0A4B00000000 + int(os.environ.get("BENCH_SEED_OFFSET", "0"))   # SURVEY 8(d) config-2 lane seed base (the offset: other samples of the same generator, for A/B runs)
MAX_STEPS = 1000


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=65536, help="playouts per GPU")
    ap.add_argument("--group", type=int, default=20, help="batches per group launch (oakgpu_rollout_group_dev), <= 64")
    ap.add_argument("--streams", type=int, default=2, help="groups in flight (contexts / HIP streams)")
    ap.add_argument("--playouts-per-lane", type=int, default=2,
                    help="k > 1: persistent n/k lanes per batch that refill from an atomic playout queue")
    ap.add_argument("--exchange", choices=["torch", "rccl"], default="torch",
                    help="config4: the all-gather of the per-root means through torch.distributed (RCCL underneath) or through "
                         "the library's own ncclAllGather call site (oakgpu_all_gather_dev)")
    ap.add_argument("--root-groups", type=int, default=4,
                    help="config4: independent groups a rank cuts its roots into (oak_amd.dist.RootGroups); 1 = one launch per step")
    ap.add_argument("--root-slice", type=int, default=16,
                    help="config4: turn-steps a launch advances each playout in flight by (oak_amd.dist.RootSteps: a playout is credited to "
                         "step k + (len - 1) // slice of its root, stragglers travel on a carry list); a power of two; 0 = round 4's grouped "
                         "form (every step waits for its longest playout, --root-groups)")
    ap.add_argument("--workload", choices=["all", "rollout", "leaf", "config3", "config4", "search"], default="all",
                    help="all (default) = the configs[1] headline line + `leaf` and `config3` sub-records; rollout = configs[1] "
                         "only; leaf = leaf-evals/s of the 768-256-256-256-1 net; config3 = configs[2]: one turn-step of the "
                         "whole batch + a leaf eval of every lane, every turn; search = tree search with batched leaves "
                         "(oakgpu_search), iterations/s on one random OU root")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch / rendezvous / exchange / accounting rehearsal WITHOUT a GPU: gloo backend, CPU tensors, the kernel "
                         "replaced by a stub that writes a rank-and-batch pattern (no engine, no oracle); the line says dry_run: true "
                         "and its value measures nothing.  tests/test_bench_launch.py runs `bench.py --gpus 2 --dry-run` here.")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` with no launcher around it: start the N ranks ourselves, BEFORE this process has made
        # any GPU call (torch is not even imported yet) -- never re-exec a process that touched the GPU
        raise SystemExit(self_launch(args.gpus))

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, torch, rank, world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists; --dry-run rehearses the launch and the exchange only)")
    from oak_amd import _lib
    from oak_amd import dist as oakdist
    from oak_amd.engine import Context
    # BENCH_REHEARSAL=1: every rank on GPU 0 with the gloo backend (RCCL refuses two ranks on one device) -- the REAL kernels and
    # the REAL N > 1 code path (lane sharding, per-group gather on its own stream, row check, sub-records with world > 1) on a
    # one-GPU box; its numbers mean nothing (the ranks share the card, the collective goes through the host)
    rehearsal = bool(os.environ.get("BENCH_REHEARSAL"))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    force_dist = bool(os.environ.get("BENCH_FORCE_DIST"))   # exercise the RCCL path with a single rank (tests)
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)  # nccl == RCCL on ROCm
        if dist.get_world_size() != world:
            raise SystemExit("process group has %d ranks, expected %d" % (dist.get_world_size(), world))

    if args.workload == "search":
        rec = search_workload(args, torch, dev, rank, local_rank, world, dist)
        if rank == 0:
            print(json.dumps(rec), flush=True)
    elif args.workload == "config4":
        rec = config4_workload(args, torch, dev, rank, local_rank, world, dist)
        if rank == 0:
            print(json.dumps(rec), flush=True)
    elif args.workload in ("leaf", "config3"):      # one sub-record on its own, same JSON shape as the headline line
        recs = leaf_records(args, torch, dev, rank, local_rank, world, dist, which=(args.workload,))
        if rank == 0:
            rec = recs[args.workload]
            rec["vs_baseline"] = None
            print(json.dumps(rec), flush=True)
    else:
        # default ("all"): the headline configs[1] line carries the whole BASELINE metric -- turn-steps/s of batched
        # playouts plus, as sub-records with their own timed regions, leaf-evals/s (`leaf`), configs[2] (`config3`),
        # configs[3] (`config4`: root-parallel MCTS step, STRONG scaling -- so one `--gpus N` command yields both curves)
        # and configs[4] (`config5`: tree search with network leaves + exact Nash at the root)
        out = rollout_workload(args, torch, dev, rank, local_rank, world, dist, force_dist)
        subs = {}
        if args.workload == "all":
            subs.update(leaf_records(args, torch, dev, rank, local_rank, world, dist) or {})
            for name, fn in (("config4", config4_workload), ("config5", search_workload)):
                try:
                    rec = fn(args, torch, dev, rank, local_rank, world, dist)
                except Exception as e:   # a sub-record must never cost the headline line (same code on every rank: they skip together)
                    rec = {"error": "%s: %s" % (type(e).__name__, e)}
                    print("bench.py: sub-record %s failed: %s" % (name, rec["error"]), file=sys.stderr, flush=True)
                if rank == 0:
                    subs[name] = rec
        if rank == 0 and world == 1 and not args.no_cpu_baseline:   # after every GPU timed region
            out["cpu_baseline"] = cpu_baseline(args.batch)
            if isinstance(subs.get("config4"), dict) and "error" not in subs["config4"]:
                subs["config4"]["cpu_baseline"] = cpu_baseline_config4()
        if rank == 0:
            out.update(subs)
            if rehearsal:
                out["rehearsal"] = "BENCH_REHEARSAL: all ranks on GPU 0 over gloo -- a code-path check, not a measurement"
            print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: N fresh child processes through torch.distributed.run (one rank per
    GPU, rendezvous on 127.0.0.1), started before this process has imported torch or touched the GPU; rank 0's JSON line
    goes to the inherited stdout; the exit code is the launcher's (non-zero if any rank failed).  The reference's analogue
    starts its N workers from one command too (cpp/src/generate.cc:527-536: N std::threads)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


def dry_run(args, torch, rank, world):
    """No GPU, no engine, no oracle: what runs is everything AROUND the kernel of the N > 1 headline path -- rendezvous (gloo),
    the group plan, disjoint lane seeds, ONE all-gather per group through oak_amd.dist.gather_round, the row check of the
    gather, barrier + max-over-ranks timing, the sum of turn-steps over ranks, rank 0's single JSON line.  The stub
    'kernel' writes value = rank + batch / 1024 and 100 turn-steps per playout."""
    import torch.distributed as dist
    from oak_amd import dist as oakdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = min(args.batch, 4096)
    G = max(1, min(args.group, 64))
    timed = [G] * (args.steps // G) + ([args.steps % G] if args.steps % G else [])
    values = torch.empty((G, n), dtype=torch.float32)
    gathered = torch.empty((world, G, n), dtype=torch.float32)
    seeds = [oakdist.lane_seed0(SEED0 + k * n * world, n * world, rank, world) for k in range(G)]
    my_steps = 0
    dist.barrier()
    t0 = time.perf_counter()
    for gi, count in enumerate(timed):
        for k in range(count):
            values[k].fill_(rank + (gi * G + k) / 1024.0)      # the stub
        my_steps += 100 * n * count
        oakdist.gather_round(values[:count], gathered[:, :count] if count == G else None)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    probe, _ = oakdist.gather_round(values)
    for r in range(world):     # rank r's rows of the gather are rank r's values
        assert torch.equal(probe[r], values - rank + r), "all-gather returned a different row for rank %d" % r
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    s = torch.tensor([my_steps], dtype=torch.int64)
    dist.all_reduce(s, op=dist.ReduceOp.SUM)
    sd = torch.tensor([seeds[0]], dtype=torch.int64)
    alls = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(alls, sd)
    if rank == 0:
        print(json.dumps({
            "metric": "turn-steps/s (batched playouts)", "value": int(s.item()) / float(t.item()), "unit": "turn-steps/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": float(t.item()) / max(args.steps, 1) * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic", "dry_run": True,
            "ranks_seen": dist.get_world_size(),
            "config": {"workload": "DRY RUN: stub kernel on CPU tensors over gloo -- rehearses launch, lane sharding, the per-group all-gather "
                                   "and the accounting of configs[1]; the value measures nothing", "batch_per_gpu": n,
                       "first_lane_seed_per_rank": [int(x.item()) for x in alls], "groups": len(timed), "group": G},
        }), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def rollout_workload(args, torch, dev, rank, local_rank, world, dist, force_dist=False):
    """BASELINE configs[1].  The K timed steps (= K independent 65,536-playout batches) are submitted in GROUPS of up to
    `--group` batches: one oakgpu_rollout_group_dev launch drains a whole group through one playout queue, so a group has
    ONE tail (its longest playout) instead of one per batch.  Groups alternate over `--streams` contexts (HIP streams,
    each with its own batch buffers), so with more than one group the next group's launch fills the SIMDs the previous
    group's tail leaves idle.  Returns the JSON record (rank 0; None elsewhere)."""
    import numpy as np
    from oak_amd import _lib
    from oak_amd import dist as oakdist
    from oak_amd.engine import Context
    n = args.batch
    G = max(1, min(args.group, 64))
    u8 = torch.uint8

    def plan(k):   # k steps -> group sizes
        return [G] * (k // G) + ([k % G] if k % G else [])
    timed, warm = plan(args.steps), plan(args.warmup)
    S = max(1, min(args.streams, len(timed)))
    gmax = max(timed + warm + [1])

    class Slot:
        """One group in flight: a context (HIP stream) and the buffers of its `gmax` batches."""

        def __init__(self, idx):
            self.ctx = Context(local_rank)   # owns a dedicated non-blocking HIP stream
            self.stream = torch.cuda.ExternalStream(self.ctx.stream_ptr(), device=dev)
            self.ctx.ensure_ou_pools()
            self.ctx.set_playouts_per_lane(args.playouts_per_lane)
            self.battles = torch.empty((gmax, n, 384), dtype=u8, device=dev)
            self.durations = torch.empty((gmax, n, 8), dtype=u8, device=dev)
            self.prng = torch.empty((gmax, n, 8), dtype=u8, device=dev)
            self.results_in = torch.empty((gmax, n), dtype=u8, device=dev)
            self.results = torch.empty((gmax, n), dtype=u8, device=dev)
            self.steps_out = torch.zeros((gmax, n), dtype=torch.int32, device=dev)
            self.values = torch.empty((gmax, n), dtype=torch.float32, device=dev)
            self.total = torch.zeros((), dtype=torch.int64, device=dev)
            self.gathered = torch.empty((world, gmax, n), dtype=torch.float32, device=dev) if exchange else None
            self.gather_done = None
            self.descs = (_lib.RolloutBatch * gmax)()
            for k in range(gmax):
                # synthetic input, generated on device; lane seeds disjoint across ranks, slots and batches
                seed0 = oakdist.lane_seed0(SEED0 + (idx * gmax + k) * n * world, n * world, rank, world)
                _lib.check(self.ctx.lib.oakgpu_random_ou_battles_dev(
                    self.ctx.handle, C.c_uint64(seed0), n, C.c_void_p(self.battles[k].data_ptr()), C.c_void_p(self.durations[k].data_ptr()),
                    C.c_void_p(self.prng[k].data_ptr()), C.c_void_p(self.results_in[k].data_ptr())))
                self.descs[k] = _lib.RolloutBatch(self.battles[k].data_ptr(), self.durations[k].data_ptr(), self.results_in[k].data_ptr(),
                                                  self.prng[k].data_ptr(), n, self.results[k].data_ptr(), self.steps_out[k].data_ptr(),
                                                  self.values[k].data_ptr(), None, None)

        def run(self, count):
            if self.gather_done is not None:          # the previous gather of this slot still reads self.values
                self.stream.wait_event(self.gather_done)
            _lib.check(self.ctx.lib.oakgpu_rollout_group_dev(self.ctx.handle, self.descs, count, MAX_STEPS, 0))

        def finish(self, count):
            with torch.cuda.stream(self.stream):
                self.total += self.steps_out[:count].sum(dtype=torch.int64)   # one small reduction kernel, inside the timed region
            if exchange:
                # the path's single exchange: ONE RCCL all-gather of the group's fp32 leaf values to every rank, on its
                # own stream behind the group's launch; only the slot's NEXT group waits for it (buffer reuse)
                ev = torch.cuda.Event()
                ev.record(self.stream)
                xstream.wait_event(ev)
                with torch.cuda.stream(xstream):
                    oakdist.gather_round(self.values[:count], self.gathered[:, :count] if count == gmax else None)
                    self.gather_done = torch.cuda.Event()
                    self.gather_done.record(xstream)

    exchange = world > 1 or force_dist
    xstream = torch.cuda.Stream(device=dev) if exchange else None
    slots = [Slot(i) for i in range(S)]
    torch.cuda.synchronize(dev)
    for sl in slots:   # setup, not warm-up: every slot's first launch allocates its context's tables and loads torch's
        sl.run(gmax)   # lazily-loaded reduce kernels (and the collective registers its buffers) -- none of that is timed
        sl.finish(gmax)
    torch.cuda.synchronize(dev)
    for k, count in enumerate(warm):
        slots[k % S].run(count)
        slots[k % S].finish(count)
    torch.cuda.synchronize(dev)
    for sl in slots:
        sl.total.zero_()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in timed]
    for k, (a, b) in enumerate(ev):   # force event creation outside the timed region
        a.record(slots[k % S].stream)
        b.record(slots[k % S].stream)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k, count in enumerate(timed):
        sl = slots[k % S]
        ev[k][0].record(sl.stream)
        sl.run(count)
        ev[k][1].record(sl.stream)
        sl.finish(count)
    torch.cuda.synchronize(dev)      # includes every gather of the timed region (xstream)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    total_steps = sum(sl.total for sl in slots)
    my_steps = int(total_steps.item())
    kern_ms = [a.elapsed_time(b) for a, b in ev]
    # the queue kernel's control words after the timed region (outside it: the call synchronises): [40] donations, [41] adoptions of the
    # last launch of each context; [63] the STICKY error word of the in-launch migration (a bounded wait that ran out) -- must be 0
    qc = [sl.ctx.queue_counters() for sl in slots]
    queue_counters = {"error_word_63": [int(c[63]) for c in qc], "donations_last_launch": [int(c[40]) for c in qc],
                      "adoptions_last_launch": [int(c[41]) for c in qc]}
    assert not any(queue_counters["error_word_63"]), "k_rollout_queue: a bounded wait of the in-launch migration ran out"
    if exchange and world > 1:       # one-off check of the exchange: rank r's rows of the last full gather are rank r's values
        sl = slots[0]
        mine = sl.values.clone()
        probe = torch.empty((world, gmax, n), dtype=torch.float32, device=dev)
        oakdist.gather_round(mine, probe)
        assert torch.equal(probe[rank], mine), "all-gather returned a different row for this rank"
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s = torch.tensor([my_steps], dtype=torch.int64, device=dev)
        dist.all_reduce(s, op=dist.ReduceOp.SUM)
        all_steps = int(s.item())
    else:
        all_steps = my_steps
    if rank != 0:
        return None

    value = all_steps / elapsed
    # roofline of the dominant kernel: one launch = one group; algorithmic bytes = 802 B x the turn-steps the launch
    # executed; duration = HIP events around the launch on its own stream
    launch_s = sum(kern_ms) / 1e3
    achieved = my_steps * ALGO_BYTES_PER_STEP / launch_s / 1e9
    # NOT measured in this run: PMC counters need rocprofv3 around the process.  The committed profile's per-turn-step figures
    # (profiles/traffic.json, written by tools/summarize_profile_r03.py from the --pmc passes) scaled by THIS run's turn-steps
    tj = profile_json()
    per_step = tj.get("k_rollout_hbm_bytes_per_turn_step")
    traffic = per_step * my_steps / len(timed) if per_step else None
    valu = tj.get("valu_wave_insts_per_turn_step")
    out = {
        "metric": "turn-steps/s (batched playouts)",
        "value": value,
        "unit": "turn-steps/s",
        "n_gpus": world,
        "ranks_seen": (dist.get_world_size() if (world > 1 or force_dist) else 1),
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u16",
        "data": "synthetic",
        "config": {
            "workload": "configs[1]: batch=65536 random OU team pairs per GPU, pure random-policy rollout to "
                        "terminal (cap 1000 turn-steps)",
            "batch_per_gpu": n,
            "playouts_per_s": n * world * args.steps / elapsed,
            "mean_turn_steps_per_playout": all_steps / (n * world * args.steps),
            "parallelism": ("lanes sharded by rank; one RCCL all-gather of the fp32 leaf values per group; " if world > 1 else "single GPU; ") +
                           "steps submitted as %d group launch(es) of up to %d batches over %d HIP stream(s)" % (len(timed), G, S),
            "group": G, "groups": len(timed), "streams": S,
            "playouts_per_lane": args.playouts_per_lane,
            "queue_counters": queue_counters,
            "fast_forward": "a PROVEN frozen standstill (both actives frozen, nobody able to act) is taken to its last turn-step in one go and its "
                            "skipped turn-steps are counted in `value` (exact: tests/test_gpu_parity.py::test_frozen_standstill_skip_is_exact; <= 0.004 % "
                            "of the count)",
            "parity": "bit-exact vs this repo's CPU oracle (libpkmn parity unpinned, see DESIGN.md)",
        },
        "roofline": {
            "bound": "hbm",
            "kernel": "oak::k_rollout_queue (one launch = one group of batches)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_source": (PROFILE_SOURCE + ": k_rollout_hbm_bytes_per_turn_step (2 x FETCH_SIZE + WRITE_SIZE of a profiled group launch) x "
                               "this run's turn-steps per launch; not measured in this run") if traffic else None,
            "avg_launch_ms": launch_s / len(timed) * 1e3,
            "launches": len(timed),
            "algorithmic_bytes_per_turn_step": ALGO_BYTES_PER_STEP,
            "turn_steps_per_launch": my_steps / len(timed),
        },
    }
    if valu:
        # the kernel is integer-VALU bound, not HBM bound: wave-instructions issued per turn-step (PMC
        # SQ_INSTS_VALU, committed in profiles/) against the MEASURED issue peak of a mixed integer instruction stream
        # (profiles/r04_valu_issue.json, tools/experiments/valu_issue_bench.hip: 4 cycles per wave64 instruction per SIMD;
        # the 2 cycles of the microarchitecture guide hold for homogeneous v_add / v_and / v_mov streams only)
        peak, peak_src = valu_issue_peak()
        ach = valu * my_steps / elapsed
        out["roofline"]["valu_issue"] = {"wave_insts_per_turn_step": valu, "achieved_ginst_s": ach / 1e9,
                                         "peak_ginst_s": peak / 1e9, "peak_source": peak_src, "frac": ach / peak,
                                         "active_lanes_per_wave_inst": tj.get("valu_active_lanes_per_wave_inst"),
                                         "source": PROFILE_SOURCE + ": SQ_INSTS_VALU per turn-step of a profiled group launch x this run's turn-steps/s"}
    # (the CPU baseline is timed by the caller AFTER every GPU record: 10-30 s of host work between two GPU timed regions let
    # the device clock down, and the next region then measured the ramp)
    return out


def leaf_records(args, torch, dev, rank, local_rank, world, dist, which=("leaf", "config3")):
    """Second metric of BASELINE.json (leaf-evals/s) and configs[2], as sub-records of the one JSON line.
    leaf    : one step = value_inference over one batch of 65,536 mid-game states (random OU battles advanced 20 random
              turn-steps on the device) with the config-3 network (768 -> 256 -> 256 -> 256 -> 1, seeded synthetic weights).
    config3 : one step = one random turn-step of the resident 65,536-lane batch, in place, + value_inference of every
              lane (BASELINE configs[2]: "MLP leaf eval every turn"); 40-turn episodes.
    Roofline: every layer priced against the dense peak of the matrix pipe it runs on (fp32 MFMA 157.3 TFLOP/s; bf16 MFMA 2,500
    TFLOP/s for the layers computed as bf16 triples, six partial products per multiply-add), one time-weighted fraction per record
    (`priced`); algorithmic FLOP per leaf from SURVEY 8(d) (oak_amd.netfile.flops_per_leaf)."""
    import tempfile
    from oak_amd import _lib, netfile
    from oak_amd import dist as oakdist
    from oak_amd.engine import Context, Network
    n = args.batch
    ctx = Context(local_rank)
    stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=dev)
    ctx.ensure_ou_pools()
    u8 = torch.uint8

    def P(t):
        return C.c_void_p(t.data_ptr())
    battles = torch.empty((n, 384), dtype=u8, device=dev)
    mid = torch.empty((n, 384), dtype=u8, device=dev)
    durations = torch.empty((n, 8), dtype=u8, device=dev)
    dur_mid = torch.empty((n, 8), dtype=u8, device=dev)
    prng = torch.empty((n, 8), dtype=u8, device=dev)
    rin = torch.empty((n,), dtype=u8, device=dev)
    rout = torch.empty((n,), dtype=u8, device=dev)
    steps_out = torch.empty((n,), dtype=torch.int32, device=dev)
    values = torch.empty((n,), dtype=torch.float32, device=dev)
    lib, h = ctx.lib, ctx.handle
    seed0 = oakdist.lane_seed0(SEED0, n * world, rank, world)
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(seed0), n, P(battles), P(durations), P(prng), P(rin)))
    ctx.synchronize()   # the generator ran on the context's stream; torch copies below run on torch's
    b0, d0, p0, r0 = battles.clone(), durations.clone(), prng.clone(), rin.clone()   # turn-0 batch (config3 episodes restart from it)
    torch.cuda.synchronize(dev)
    _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 20, 0, P(rout), P(steps_out),
                                      P(values), P(mid), P(dur_mid)))
    ctx.synchronize()
    td = tempfile.mkdtemp()
    path = os.path.join(td, "config3.battle.net")
    netfile.write_random_net(path, seed=7, hidden=256, value_hidden=256)
    net = Network(ctx, path=path)
    main_f, emb_f = netfile.flops_per_leaf(256, 256)
    live = torch.zeros((n,), dtype=torch.int32, device=dev)   # turn-steps done per lane (summed after the timed region)
    turn = [0]
    # configs[2] evaluates the SAME resident lanes every turn: the party-slot embeddings are cached by exact identity tags
    # (oakgpu_leaf_eval_cached_dev, the GPU form of the reference's PokemonCache) -- buffers the caller keeps between calls
    emb = torch.empty((n, 768), dtype=torch.float32, device=dev)
    tags = torch.full((n, 10, 6), -1, dtype=torch.int32, device=dev)

    def leaf_step():
        _lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, P(mid), P(dur_mid), n, P(values), None))

    def config3_step():
        if turn[0] % 40 == 0:   # restoring the turn-0 batch is a 26 MB device copy
            with torch.cuda.stream(stream):
                battles.copy_(b0); durations.copy_(d0); prng.copy_(p0); rin.copy_(r0)
        turn[0] += 1
        _lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 1, 0, P(rin), P(steps_out),
                                          P(values), P(battles), P(durations)))
        _lib.check(lib.oakgpu_leaf_eval_cached_dev(h, net.handle, P(battles), P(durations), n, P(values), P(emb), P(tags)))
        with torch.cuda.stream(stream):
            live.add_(steps_out)

    def timed(step, K, W):
        for _ in range(max(W, 1)):
            step()
        turn[0] = 0
        with torch.cuda.stream(stream):
            live.zero_()
        ctx.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        for a, b in ev:
            a.record(stream)
            b.record(stream)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for k in range(K):
            ev[k][0].record(stream)
            step()
            ev[k][1].record(stream)
            if world > 1:   # the path's exchange: the batch's leaf values to every rank
                with torch.cuda.stream(stream):
                    oakdist.gather_values(values, n * world)
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, sum(a.elapsed_time(b) for a, b in ev) / K / 1e3

    def kernel_us():   # diagnostic pass outside the timed regions: HIP events around each kernel of one call
        _lib.check(lib.oakgpu_set_kernel_timing(h, 1))
        acc = [0.0, 0.0, 0.0]
        for _ in range(5):
            leaf_step()
            ms = (C.c_float * 3)()
            _lib.check(lib.oakgpu_get_leaf_kernel_ms(h, ms))
            acc = [x + y for x, y in zip(acc, ms)]
        _lib.check(lib.oakgpu_set_kernel_timing(h, 0))
        return {"k_embed_prows (party slots)": acc[0] / 5 * 1e3, "k_embed_arows (actives)": acc[1] / 5 * 1e3, MAIN_KERNEL: acc[2] / 5 * 1e3}

    # the main net's kernel: fp32 values as scaled fp16 pairs on the fp16 matrix pipe (k_mainnet_pair, the default since round 5) unless
    # OAKGPU_MAIN_NET=bf16x3 (bf16 triples, k_mainnet_split) or =fp32 (fp32 MFMA, k_mainnet_wave); include/oakgpu.h
    main_env = os.environ.get("OAKGPU_MAIN_NET", "")
    main_mode = "fp32" if main_env == "fp32" else "split" if main_env == "bf16x3" else "pair"
    split = main_mode != "fp32"
    MAIN_KERNEL = {"pair": "k_mainnet_pair<8>", "split": "k_mainnet_split<8>", "fp32": "k_mainnet_wave"}[main_mode]
    MAIN_PRODUCTS = {"pair": 3, "split": 6, "fp32": 1}[main_mode]   # matrix-pipe products executed per algorithmic multiply-add
    EMB = ("fp32 results throughout.  Embedding nets: one-hot rows summed in fp32 on the vector ALUs; dense features, the move of the sums into the item "
           "lanes and the second layers as exact bf16 triples on the bf16 pipe.  ")
    ARITH = {"pair": EMB + "Main net: every fp32 value times an exact power of two (one per weight row = output feature, one per batch row for the activations) is the sum of "
                           "two round-to-nearest fp16 parts (to 2^-24), and every product runs as its three largest fp16 x fp16 partial products (exact in the fp32 "
                           "accumulator; dropped: < 2^-24 of the product) on v_mfma_f32_32x32x16_f16, fp32 accumulation; error vs float64 at the fp32-MFMA "
                           "kernel's level (tests/test_gpu_leafnet.py::test_pair_and_triple_main_nets_are_fp32_results)",
             "split": EMB + "Main net: every fp32 value is the exact sum of three bf16 parts and every product runs as its six largest bf16 x bf16 partial products "
                            "(exact in the fp32 accumulator; dropped: < 2^-24 of the product) on v_mfma_f32_32x32x16_bf16, fp32 accumulation",
             "fp32": "embedding nets: first layers fp32 vector ALUs, second layers bf16 triples; main net fp32 MFMA"}[main_mode]

    def policy_note(value_call_s):   # SURVEY 8 row f3, untimed diagnostic pass: value_policy_inference (network.h:102-123) over the same states
        c1, c2 = (torch.empty((n, 9), dtype=u8, device=dev) for _ in range(2))
        n1, n2 = (torch.empty((n,), dtype=u8, device=dev) for _ in range(2))
        l1, l2 = (torch.empty((n, 9), dtype=torch.float32, device=dev) for _ in range(2))
        _lib.check(lib.oakgpu_choices_dev(h, P(mid), P(rout), 0, P(c1), P(n1), n))
        _lib.check(lib.oakgpu_choices_dev(h, P(mid), P(rout), 1, P(c2), P(n2), n))

        def call():
            _lib.check(lib.oakgpu_leaf_eval_policy_dev(h, net.handle, P(mid), P(dur_mid), n, P(c1), P(n1), P(c2), P(n2), P(values), P(l1), P(l2)))
        for _ in range(5):
            call()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ctx.synchronize()
        a.record(stream)
        for _ in range(10):
            call()
        b.record(stream)
        ctx.synchronize()
        t = a.elapsed_time(b) / 10 / 1e3
        return {"what": "value_policy_inference: the value + the logits of every legal choice of both sides (<= 9 each), 64-wide policy heads",
                "leaf_evals_per_s": n / t, "ms_per_call": t * 1e3, "policy_heads_ms": (t - value_call_s) * 1e3,
                "kernel": "the leaf call's kernels + oak::k_policy_rows<2>", "parity": "<= 2e-5 per logit vs the numpy oracle (tests/test_gpu_leafnet.py)"}

    def leaf_kernels(kus, party_fraction=1.0):
        """The call's kernels priced per pipe (SURVEY 8d's per-leaf FLOP split by layer): first layers of the embedding nets against
        the fp32 peak (one-hot / dense rows: nnz x 128 multiply-adds per item; since round 4 the row sums run on the vector ALUs,
        whose fp32 peak equals the fp32 matrix pipe's 157.3 TFLOP/s -- the fp32 MFMAs they replaced ran on those ALUs anyway:
        tools/experiments/mfma_overlap_bench.hip), their second layers and the main net's three
        dense layers as bf16 triples on the bf16 pipe (6 partial products per multiply-add), the 256 -> 1 head on the vector pipe
        (not priced: 512 FLOP per leaf).  party_fraction: the share of the party slots actually re-embedded (configs[2]'s cache)."""
        p1, p2 = 10 * 2 * 12 * 128 * party_fraction, 10 * 2 * 128 * 59 * party_fraction
        a1, a2 = 2 * 2 * 45 * 128, 2 * 2 * 128 * 83
        mainf = main_f - 2 * 256
        F, B = FP32_MATRIX_TFLOPS * 1e12, BF16_MATRIX_TFLOPS * 1e12
        ks = [{"name": "oak::k_embed_prows (party slots; half of oak::k_embed_both)", "us": kus["k_embed_prows (party slots)"],
               "parts": [("first layer (6 dense + 7 one-hot rows per item)", "fp32 vector ALUs (157.3 TFLOP/s, = the fp32 MFMA peak)", p1 * n, F), ("second layer 128 -> 59 as bf16 triples", "bf16 MFMA x 6", 6 * p2 * n, B)]},
              {"name": "oak::k_embed_arows (actives; the other half)", "us": kus["k_embed_arows (actives)"],
               "parts": [("first layer (36 dense + 17 one-hot / move rows per item)", "fp32 vector ALUs (157.3 TFLOP/s, = the fp32 MFMA peak)", a1 * n, F), ("second layer 128 -> 83 as bf16 triples", "bf16 MFMA x 6", 6 * a2 * n, B)]}]
        if split:   # (the fp16 and the bf16 matrix pipes have the same dense peak)
            ks.append({"name": "oak::" + MAIN_KERNEL, "us": kus[MAIN_KERNEL],
                       "parts": [("768 -> 256 -> 256 -> 256 as " + ("scaled fp16 pairs" if main_mode == "pair" else "bf16 triples"),
                                  "%s MFMA x %d" % ("fp16" if main_mode == "pair" else "bf16", MAIN_PRODUCTS), MAIN_PRODUCTS * mainf * n, B)]})
        else:
            ks.append({"name": "oak::" + MAIN_KERNEL, "us": kus[MAIN_KERNEL], "parts": [("768 -> 256 -> 256 -> 256", "fp32 MFMA", mainf * n, F)]})
        return ks

    out = {}
    tj = profile_json()
    # sub-records: at least 30 untimed calls (~20 ms) before the timed K -- the clocks settle over the first ~15 ms of MFMA
    # work after the integer-VALU rollout phase, and with the driver's --warmup 5 the 14 ms timed region measured that ramp
    K, W = args.steps, max(args.warmup, 30)
    if "leaf" in which:
        elapsed, avg_s = timed(leaf_step, K, W)
        kus = kernel_us()
        frac, rows, tmin, tsum = priced(leaf_kernels(kus))
        achieved = (main_f + emb_f) * n / avg_s / 1e12
        rec = {
            "metric": "leaf-evals/s", "value": n * world * K / elapsed, "unit": "leaf-evals/s", "n_gpus": world, "steps": K,
            "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "dtype": "f32",
            "data": "synthetic",
            "note": "warmup: at least 30 untimed calls in front of the timed ones whatever --warmup says -- the clocks settle over the first ~15 ms of matrix "
                    "work after the integer rollout phase, and with the driver's --warmup 5 the timed region measured that ramp",
            "config": {"workload": "leaf part of configs[2]: value_inference (encode + embeddings + 768-256-256-256-1 MainNet + "
                                   "sigmoid) over 65536 mid-game states per GPU", "batch_per_gpu": n,
                       "parity": "<= 1e-5 vs numpy oracle pinned by the reference torch mirror"},
            "arithmetic": ARITH,
            # `achieved` / `peak` / `frac`: ALGORITHMIC fp32 FLOP per second of the whole call, and what that rate would be with every
            # layer at the peak of the pipe it runs on (time-weighted over the call's kernels) -- frac = achieved / peak <= 1
            "roofline": {"bound": "mfma", "kernel": "oak::k_embed_both (k_embed_prows + k_embed_arows in one launch) + oak::%s (one value_inference call)" % MAIN_KERNEL,
                         "achieved": achieved, "peak": achieved / (tmin / avg_s), "unit": "TFLOP/s", "frac": tmin / avg_s,
                         "frac_note": "sum over the call's layers of (algorithmic work on its pipe / that pipe's dense peak: fp32 (vector ALUs = fp32 MFMA) 157.3 TFLOP/s, bf16 / fp16 MFMA "
                                      "2,500 TFLOP/s with 6 (bf16 triples) or 3 (fp16 pairs) partial products per multiply-add) / the call's measured time; `kernels` has it per kernel "
                                      "(durations from HIP events around each launch, a diagnostic pass outside the timed region: "
                                      "time-weighted %.3f over those)" % frac,
                         "kernels": rows,
                         "traffic": (tj.get("leaf_hbm_bytes_per_leaf") * n if tj.get("leaf_hbm_bytes_per_leaf") else None),
                         "traffic_source": (PROFILE_SOURCE + ": leaf_hbm_bytes_per_leaf (2 x FETCH_SIZE + WRITE_SIZE of the call's kernels) x batch; not measured in this run")
                         if tj.get("leaf_hbm_bytes_per_leaf") else None,
                         "avg_call_ms": avg_s * 1e3, "algorithmic_flop_per_leaf": main_f + emb_f, "mainnet_flop_per_leaf": main_f,
                         "kernel_us": kus},
        }
        rec["policy"] = policy_note(avg_s)
        out["leaf"] = rec
    if "config3" in which:
        elapsed, avg_s = timed(config3_step, K, W)
        achieved = (main_f + emb_f) * n / avg_s / 1e12
        steps_done = int(live.sum(dtype=torch.int64).item())
        # what the cache actually executes: a diagnostic pass (untimed) reads the work list's length after every turn of one episode
        miss = []
        turn[0] = 0
        for _ in range(40):
            config3_step()
            cnt = C.c_uint32(0)
            _lib.check(lib.oakgpu_leaf_cache_last_count(h, C.byref(cnt)))
            miss.append(cnt.value / (n * 10.0))
        miss_rate = sum(miss) / len(miss)
        party_f = 10 * 2 * (12 * 128 + 128 * 59)
        executed = (main_f + (emb_f - party_f) + party_f * miss_rate) * n / avg_s / 1e12
        # the step priced per pipe: the turn-step and the tag comparison against the HBM roof with their algorithmic bytes (802 B per
        # turn-step, 480 B of tags per leaf), the leaf kernels as in `leaf` with the party pass scaled to the slots the cache re-embeds
        lk = leaf_kernels(out["leaf"]["roofline"]["kernel_us"] if "leaf" in out else kernel_us(), miss_rate)
        tmin_leaf = sum(w / pk for k_ in lk for _, _, w, pk in k_["parts"])
        tmin_hbm = (ALGO_BYTES_PER_STEP + 480) * n / (HBM_PEAK_GBPS * 1e9)
        frac3 = (tmin_leaf + tmin_hbm) / avg_s
        if world > 1:
            s_ = torch.tensor([steps_done], dtype=torch.int64, device=dev)
            dist.all_reduce(s_, op=dist.ReduceOp.SUM)
            steps_done = int(s_.item())
        rec = {
            "metric": "turn-steps/s (rollout + leaf eval every turn)", "value": steps_done / elapsed, "unit": "turn-steps/s",
            "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True,
            "scaling": "weak", "dtype": "f32", "data": "synthetic", "arithmetic": ARITH,
            "note": "warmup: at least 30 untimed steps (see `leaf`)",
            "config": {"workload": "configs[2]: batch=65536 random OU team pairs per GPU; every step = one random turn-step of the "
                                   "whole batch (in place) + value_inference (768-256-256-256-1) of every lane, party-slot embeddings cached by identity "
                                   "tags (PokemonCache analogue); 40-turn episodes",
                       "batch_per_gpu": n, "leaf_evals_per_s": n * world * K / elapsed,
                       "live_lane_fraction": steps_done / (n * world * K)},
            "roofline": {"bound": "mfma", "kernel": "oak::k_rollout_staged (1 turn-step) + oak::k_party_tags + oak::k_embed_both<list> (changed party slots + "
                                                     "actives) + oak::" + MAIN_KERNEL,
                         "achieved": executed, "peak": executed / frac3, "unit": "TFLOP/s", "frac": frac3,
                         "frac_note": "EXECUTED work per pipe over the step's measured time: the matrix layers as in `leaf` (the party pass scaled by "
                                      "party_slot_miss_rate: the cache re-embeds only the changed slots), the turn-step and the tag pass as 802 + 480 "
                                      "algorithmic bytes per lane against the 8 TB/s HBM roof; `achieved` = executed fp32-equivalent TFLOP/s",
                         "algorithmic_tflops_if_every_embedding_were_recomputed": achieved, "party_slot_miss_rate": miss_rate,
                         "us_at_peak": {"matrix_layers": tmin_leaf * 1e6, "turn_step_and_tags_hbm": tmin_hbm * 1e6},
                         "traffic": (tj.get("config3_hbm_bytes_per_lane_turn") * n if tj.get("config3_hbm_bytes_per_lane_turn") else None),
                         "traffic_source": (PROFILE_SOURCE + ": config3_hbm_bytes_per_lane_turn x batch; not measured in this run")
                         if tj.get("config3_hbm_bytes_per_lane_turn") else None,
                         "avg_step_ms": avg_s * 1e3, "algorithmic_flop_per_leaf": main_f + emb_f},
        }
        out["config3"] = rec
    if rank == 0 and world == 1 and not args.no_cpu_baseline:   # CPU baselines after both GPU timed regions (see main)
        if "leaf" in out:
            out["leaf"]["cpu_baseline"] = cpu_baseline_leaf(path, mid, dur_mid)
        if "config3" in out:
            out["config3"]["cpu_baseline"] = cpu_baseline_config3(path, n)
    net.close()
    ctx.close()
    return out


def config4_workload(args, torch, dev, rank, local_rank, world, dist):
    """BASELINE configs[3]: root-parallel MCTS, 256 roots x 4096 playouts per search step, STRONG scaling: the roots are
    sharded contiguous-by-root over the ranks (oak_amd.dist.root_shard); every rank cuts ITS roots into independent GROUPS
    (oak_amd.dist.RootGroups, `--root-groups`): a group's step = root prep (battle.rng from the lane's stream +
    randomize_hidden_variables, mcts.h:250-263) + the playouts of its roots -> one mean per root on the device
    (oakgpu_segment_mean_dev) -> ONE all-gather of the group's means -> means on the HOST.  The roots are independent trees
    (the reference's workers never wait for each other, generate.cc:527-536), so a group's next step starts as soon as ITS
    means have arrived, whatever the other groups are doing: one group's bulk fills the SIMDs another group's tail -- its
    1,000-step playouts -- leaves idle.  Every root performs exactly K steps; per-root results do not depend on the grouping
    (tests/test_gpu_parity.py::test_root_groups_pipeline_gives_the_unpipelined_per_root_results).
    At N = 1 the record also carries `rank_share`: the SAME pipelined code over 32 roots (one rank's share at 8 GPUs), and the
    8-GPU speed-up that projects -- a one-GPU measurement of a rank's work, NOT a hardware scaling curve."""
    import numpy as np
    from oak_amd import _lib
    from oak_amd import dist as oakdist
    from oak_amd.engine import Context
    n_roots, reps = 256, 4096
    ctx = Context(local_rank)
    ctx.ensure_ou_pools()
    lib, h = ctx.lib, ctx.handle
    u8 = torch.uint8

    def P(t):
        return C.c_void_p(t.data_ptr())
    # the 256 roots = the first 256 lanes of config 2 after update(0, 0), identical on every rank
    rb, rd, rp, rr = (torch.empty(s_, dtype=u8, device=dev) for s_ in ((n_roots, 384), (n_roots, 8), (n_roots, 8), (n_roots,)))
    _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(SEED0), n_roots, P(rb), P(rd), P(rp), P(rr)))
    ctx.synchronize()
    comm = [None]
    if args.exchange == "rccl":      # the library's own ncclAllGather call site; the id travels through torch.distributed
        idt = torch.zeros(128, dtype=u8)
        if rank == 0:
            buf = (C.c_uint8 * 128)()
            _lib.check(lib.oakgpu_comm_unique_id(buf))
            idt = torch.tensor(list(buf), dtype=u8)
        if world > 1:
            idt = idt.to(dev)
            dist.broadcast(idt, 0)
            idt = idt.cpu()
        idb = (C.c_uint8 * 128)(*idt.tolist())
        comm[0] = C.c_void_p()
        _lib.check(lib.oakgpu_comm_create(h, idb, rank, world, C.byref(comm[0])))

    rccl = args.exchange == "rccl"    # the library's communicator lives on `ctx`'s stream: ONE group, on that context

    def make_context():
        if rccl:
            return ctx
        c = Context(local_rank)
        c.ensure_ou_pools()
        # several launches share the device: one lane per playout and no spreading over idle wave slots (measured,
        # profiles/r04_config4_pipeline.json: 256 roots in 4 groups 12.5 ms per step against 14.2 ms as one queue launch)
        c.set_playouts_per_lane(1)
        _lib.check(c.lib.oakgpu_set_spread(c.handle, 0))
        return c

    def run(lo, hi, groups, K, W, w, r, exchange_on):
        """K timed steps of roots [lo, hi) in `groups` groups as rank r of w; returns (seconds, my turn-steps, groups used, last means)."""
        mine = hi - lo
        n = mine * reps
        battles = rb[lo:hi].repeat_interleave(reps, 0).contiguous()
        durations = rd[lo:hi].repeat_interleave(reps, 0).contiguous()
        rin = rr[lo:hi].repeat_interleave(reps, 0).contiguous()
        # one fast_prng stream per replica, seeded by its GLOBAL lane index (results do not depend on the number of ranks or groups)
        prng = torch.empty((n, 8), dtype=u8, device=dev)
        tb, tdur, tr = torch.empty((n, 384), dtype=u8, device=dev), torch.empty((n, 8), dtype=u8, device=dev), torch.empty((n,), dtype=u8, device=dev)
        _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0xC40000000000 + lo * reps), n, P(tb), P(tdur), P(prng), P(tr)))
        ctx.synchronize()
        del tb, tdur, tr
        G = max(1, min(groups, (n_roots // w) if exchange_on else mine))
        per = oakdist.group_padding(n_roots, w, G) if exchange_on else None

        def exchange(means, out):     # on the group's stream
            if rccl:
                _lib.check(lib.oakgpu_all_gather_dev(h, comm[0], P(means), P(out), per))
            else:
                dist.all_gather_into_tensor(out, means)
        ex = exchange if (exchange_on and (w > 1 or rccl)) else None
        rg = oakdist.RootGroups(make_context, dev, battles, durations, rin, prng, mine, reps, G, world=(w if exchange_on else 1), exchange=ex, per=per,
                                owns_contexts=not rccl)
        ordered = exchange_on and w > 1
        rg.run(max(W, 1), ordered=ordered)
        rg.total.zero_()
        torch.cuda.synchronize(dev)
        if exchange_on and w > 1:
            dist.barrier()
        t0 = time.perf_counter()
        rg.run(K, ordered=ordered, keep=True)
        torch.cuda.synchronize(dev)
        if exchange_on and w > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        my_steps = int(rg.total.sum().item())
        blocks = [g_["history"][-1] for g_ in rg.groups]
        rg.close()
        del battles, durations, rin, prng
        return dt, my_steps, G, blocks

    def run_sliced(lo, hi, K, W, w, r, exchange_on):
        """K timed search steps of roots [lo, hi) in slices (oak_amd.dist.RootSteps) as rank r of w: a step's launch advances every playout
        in flight by <= --root-slice turn-steps, credits the ones that finished to this step and carries the rest; the step is over when
        its per-root aggregates (of every rank) are on the host, and only then is the next step launched.  Returns (seconds, turn-steps
        executed by the timed launches, last step's gathered record, carried playouts after the last step)."""
        mine = hi - lo
        n = mine * reps
        prng = torch.empty((n, 8), dtype=u8, device=dev)
        tb, tdur, tr = torch.empty((n, 384), dtype=u8, device=dev), torch.empty((n, 8), dtype=u8, device=dev), torch.empty((n,), dtype=u8, device=dev)
        # one fast_prng stream per (root, replica), seeded by its GLOBAL lane index (results do not depend on the number of ranks)
        _lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0xC40000000000 + lo * reps), n, P(tb), P(tdur), P(prng), P(tr)))
        ctx.synchronize()
        del tb, tdur, tr
        per = -(-n_roots // w) if exchange_on else mine

        def exchange(send, recv):     # on the context's stream: `per` int64 per rank
            if rccl:
                _lib.check(lib.oakgpu_all_gather_dev(h, comm[0], P(send), P(recv), 2 * per))      # (counted in floats: 2 per int64)
            else:
                dist.all_gather_into_tensor(recv, send)
        ex = exchange if (exchange_on and (w > 1 or rccl)) else None
        rs = oakdist.RootSteps(ctx, dev, rb[lo:hi].contiguous(), rd[lo:hi].contiguous(), rr[lo:hi].contiguous(), prng, mine, reps,
                               slice=args.root_slice, max_steps=MAX_STEPS, world=(w if exchange_on else 1), exchange=ex, per=per)
        # warm-up: the carry lists reach their steady population after ~250 / slice steps (99.5 % of the playouts end before 250 turn-steps)
        for _ in range(max(W, 2 + 256 // max(args.root_slice, 1))):
            rs.step()
            rs.finish()
        torch.cuda.synchronize(dev)
        if exchange_on and w > 1:
            dist.barrier()
        t0 = time.perf_counter()
        steps_done = 0
        for _ in range(K):
            rs.step()
            rec = rs.finish()        # the aggregates of every rank are on the host: the step is over
            steps_done += rec["turn_steps"]
        torch.cuda.synchronize(dev)
        if exchange_on and w > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        carried = rec["carried"]
        rs.close()
        del prng
        return dt, steps_done, rec, carried

    sliced = args.root_slice > 0
    if rccl:
        args.root_groups = 1
    lo, hi = oakdist.root_shard(n_roots, rank, world)
    K = args.steps
    if sliced:
        elapsed, my_steps, last, carried = run_sliced(lo, hi, K, args.warmup, world, rank, True)
        G, blocks = 1, None
    else:
        elapsed, my_steps, G, blocks = run(lo, hi, args.root_groups, K, args.warmup, world, rank, True)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        s_ = torch.tensor([my_steps], dtype=torch.int64, device=dev)
        dist.all_reduce(s_, op=dist.ReduceOp.SUM)
        all_steps = int(s_.item())
    else:
        all_steps = my_steps
    if sliced:
        per_ = -(-n_roots // world)
        cnt = oakdist.assemble_rank_blocks(n_roots, world, per_, last["count"])
        got = oakdist.credited_means(cnt, oakdist.assemble_rank_blocks(n_roots, world, per_, last["sum2"]))
        # in the steady state a step is credited as many playouts as it starts (they are other steps' stragglers, not its own)
        assert abs(int(cnt.sum()) - n_roots * reps) <= n_roots * reps // 50, "a step's credited playouts are not ~ roots x playouts"
    else:
        got = oakdist.assemble_group_means(n_roots, world, G, blocks)
    assert got.shape == (n_roots,) and ((got >= 0) & (got <= 1)).all(), "the gathered per-root means are not the 256 roots' means"
    rec = None
    if rank == 0:
        tj = profile_json()
        per_step = tj.get("config4_hbm_bytes_per_turn_step")
        rec = {
            "metric": "turn-steps/s (root-parallel MCTS step: 256 roots x 4096 playouts)", "value": all_steps / elapsed, "unit": "turn-steps/s",
            "n_gpus": world, "ranks_seen": (dist.get_world_size() if dist is not None else 1),
            "steps": K, "warmup": args.warmup, "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u16", "data": "synthetic",
            "config": {"workload": (
                           "configs[3]: root-parallel MCTS, 256 roots x 4096 fresh playouts per search step and root (root prep + rollout), roots sharded "
                           "contiguous-by-root over the ranks; a step's launch advances every playout in flight by at most %d turn-steps "
                           "(oakgpu_root_steps / k_root_step): a playout of len turn-steps started in step k is credited to step k + (len - 1) // %d of "
                           "its root -- a function of its own length, never of the schedule -- and travels between launches as a bit-exact state image "
                           "(the reference's workers never wait for each other either, generate.cc:527-536); per-root aggregates (count, 2 x value sum: "
                           "integers) folded into the kernel's retire path, ONE all-gather of them per step, the next step is launched only when they "
                           "are on the host; `value` counts the turn-steps the timed launches EXECUTED (steady state: as many playouts credited as "
                           "started per step)" % (args.root_slice, args.root_slice)) if sliced else (
                           "configs[3], round 4's form (--root-slice 0): every step runs its playouts to terminal; the rank's roots cut into independent "
                           "groups (no barrier across roots: a group's next step starts when ITS means are on the host); per-root means reduced on the "
                           "device, ONE all-gather per group and step"),
                       "roots": n_roots, "playouts_per_root": reps, "roots_per_gpu": hi - lo, "root_groups_per_gpu": G,
                       "slice_turn_steps": args.root_slice if sliced else None, "pipelined": bool(sliced or G > 1),
                       "carried_playouts_after_last_step": (carried if sliced else None),
                       "playouts_per_s": n_roots * reps * K / elapsed,
                       "exchange": ("oakgpu_all_gather_dev (ncclAllGather)" if comm[0] is not None else "torch.distributed all_gather_into_tensor, one per step"
                                    if world > 1 else "none (one rank)"),
                       "mean_root_value": float(got.mean())},
            "roofline": {"bound": "hbm", "kernel": ("oak::k_root_step (one launch = one search step of this rank's roots: carried playouts resumed, fresh ones "
                                                    "prepared, everything advanced one slice)") if sliced else
                                                   ("oak::k_rollout_regs (one launch = one search step of one group's roots, root prep included)" if G > 1 else
                                                    "oak::k_rollout_queue (one launch = one search step of this rank's roots, root prep included)"),
                         "achieved": my_steps * ALGO_BYTES_PER_STEP / elapsed / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": my_steps * ALGO_BYTES_PER_STEP / elapsed / 1e9 / HBM_PEAK_GBPS,
                         "traffic": (per_step * my_steps / K if per_step else None),
                         "traffic_source": (PROFILE_SOURCE + ": config4_hbm_bytes_per_turn_step (2 x FETCH_SIZE + WRITE_SIZE of a profiled step) x this run's "
                                            "turn-steps per step; not measured in this run") if per_step else None,
                         "algorithmic_bytes_per_turn_step": ALGO_BYTES_PER_STEP,
                         "note": "whole-job clock (launches + gathers + host copies + the host's wait for every step's aggregates), per rank.  NOTIONAL, and "
                                 "above 1 by construction: SURVEY 8d's 802 algorithmic bytes per turn-step describe a step-per-launch design; this kernel keeps "
                                 "a playout on chip for a whole slice and moves `traffic` bytes.  The bound that applies is `valu_issue`."},
        }
    if rank == 0 and sliced and tj.get("config4_valu_wave_insts_per_turn_step"):
        peak, peak_src = valu_issue_peak()
        vpt = tj["config4_valu_wave_insts_per_turn_step"]
        ach = vpt * my_steps / elapsed
        rec["roofline"]["valu_issue"] = {
            "wave_insts_per_turn_step": vpt, "achieved_ginst_s": ach / 1e9, "peak_ginst_s": peak / 1e9, "peak_source": peak_src, "frac": ach / peak,
            "active_lanes_per_wave_inst": tj.get("config4_valu_active_lanes_per_wave_inst"),
            "source": PROFILE_SOURCE + ": " + tj.get("config4_valu_source", ""),
            "note": "the peak is the best MIXED integer stream of the issue microbenchmark (4 cycles per wave64 instruction); homogeneous v_add / v_and / v_mov "
                    "runs issue at 2 cycles (1,000-1,160 G/s), and a kernel whose stream holds such runs can pass the mixed figure -- read a fraction near "
                    "or above 1 as `at the issue roof`, not as a measurement error"}
    if world == 1 and not rccl and not os.environ.get("BENCH_NO_RANK_SHARE"):
        # one rank's share at 8 GPUs (32 roots x 4096), same pipelined code, on this one GPU: what bounds the strong-scaling curve
        share = n_roots // 8
        if sliced:
            e8, s8, _, _ = run_sliced(0, share, K, max(args.warmup, 2), 1, 0, False)
            g8 = 1
        else:
            e8, s8, g8, _ = run(0, share, min(args.root_groups, 2), K, max(args.warmup, 2), 1, 0, False)
        rec["rank_share"] = {
            "what": "ONE rank's share of configs[3] at 8 GPUs (%d roots x %d playouts per step) through the same code on this one GPU" % (share, reps),
            "rank_share_ms": e8 / K * 1e3, "root_groups": g8, "slice_turn_steps": args.root_slice if sliced else None, "turn_steps_per_s": s8 / e8,
            "frac_of_full_job_rate": (s8 / e8) / (all_steps / elapsed),
            "projected_8gpu_speedup": (elapsed / K) / (e8 / K),
            "note": "projection from a one-GPU measurement, NOT a hardware curve: 8 ranks each take rank_share_ms per step (the all-gather of 256 "
                    "aggregates is microseconds), so 8 GPUs would run the job full-job-ms / rank_share_ms times faster than one."
                    + ("" if sliced else "  In this form the share's step cannot end before its longest playout -- 1,000 DEPENDENT turn-steps at a lone "
                                         "lane's ~6.5 us each."),
        }
    torch.cuda.synchronize(dev)
    if comm[0] is not None:
        lib.oakgpu_comm_destroy(comm[0])     # before the stream and buffers it used
    del rb, rd, rp, rr
    torch.cuda.empty_cache()
    ctx.close()
    return rec


def search_workload(args, torch, dev, rank, local_rank, world, dist):
    """MCTS::Search::run with batched leaves (oak_amd/csrc/search_host.hip).  One step = one search of
    2^18 iterations (batch 16,384 descents, UCB c = 2, Monte-Carlo leaves, default_search{3, 1} roll clamping) from a
    random OU turn-1 root; every rank searches its own root (root-parallel), nothing is exchanged."""
    from oak_amd import _lib
    from oak_amd import dist as oakdist
    from oak_amd.engine import Context
    from oak_amd.search import tree_search
    ctx = Context(local_rank)
    ctx.ensure_ou_pools()
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    tb, td, tp, tr = (torch.empty(s_, dtype=torch.uint8, device=dev) for s_ in ((1, 384), (1, 8), (1, 8), (1,)))

    def P(t):
        return C.c_void_p(t.data_ptr())
    seed0 = oakdist.lane_seed0(SEED0, world, rank, world)
    _lib.check(ctx.lib.oakgpu_random_ou_battles_dev(ctx.handle, C.c_uint64(seed0), 1, P(tb), P(td), P(tp), P(tr)))
    torch.cuda.synchronize(dev)
    b, d, r = tb.cpu().numpy()[0], td.cpu().numpy()[0], int(tr.cpu().numpy()[0])
    iters, batch = 1 << 18, 16384
    K = max(1, min(args.steps, 8))
    tree_search(ctx, b, d, r, iterations=2 * batch, batch=batch)      # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    outs = [tree_search(ctx, b, d, r, iterations=iters, batch=batch, seed=k) for k in range(K)]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # BASELINE configs[4]: the same search fed by batched network leaf evaluations (768-256-256-256-1) instead of rollouts,
    # exact Nash of the root matrix solved on the host (oakgpu_search_output.nash_value, csrc/nash.hpp) after every search
    import tempfile
    from oak_amd import netfile
    from oak_amd.engine import Network
    path = os.path.join(tempfile.mkdtemp(), "config5.battle.net")
    netfile.write_random_net(path, seed=7, hidden=256, value_hidden=256)
    net = Network(ctx, path=path)
    tree_search(ctx, b, d, r, iterations=2 * batch, batch=batch, evaluator=net)      # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    outs_nn = [tree_search(ctx, b, d, r, iterations=iters, batch=batch, seed=k, evaluator=net) for k in range(K)]
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed_nn = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed_nn], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_nn = float(t.item())
    single_rate = iters * K * world / elapsed_nn
    # ... and R roots per GPU at once (oakgpu_search_many): one tree per root -- the positions of R self-play games -- each on its own
    # context and its share of the host cores; a single tree leaves the card mostly idle (its walk is host work).  Every search is
    # the search it would be alone (tests/test_gpu_search.py::test_concurrent_searches_equal_the_searches_run_alone).
    from oak_amd.search import tree_search_many
    R = int(os.environ.get("BENCH_SEARCH_ROOTS", "8"))
    cores, _src = host_threads()
    os.environ["OAKGPU_SEARCH_CORES"] = str(cores)           # (the job's CPU share, not the host's affinity mask)
    rtb, rtd, rtp, rtr = (torch.empty(s_, dtype=torch.uint8, device=dev) for s_ in ((R, 384), (R, 8), (R, 8), (R,)))
    _lib.check(ctx.lib.oakgpu_random_ou_battles_dev(ctx.handle, C.c_uint64(oakdist.lane_seed0(SEED0 + 4096, R * world, rank, world)), R, P(rtb), P(rtd), P(rtp), P(rtr)))
    torch.cuda.synchronize(dev)
    rb_, rd_, rr_ = rtb.cpu().numpy(), rtd.cpu().numpy(), rtr.cpu().numpy()
    many_ctx = [Context(local_rank) for _ in range(R)]
    tree_search_many(many_ctx, rb_, rd_, rr_, list(range(R)), iterations=2 * batch, batch=batch, evaluator=net)      # warm-up
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for k in range(K):
        outs_many = tree_search_many(many_ctx, rb_, rd_, rr_, [100 * k + i for i in range(R)], iterations=iters, batch=batch, evaluator=net)
    torch.cuda.synchronize(dev)
    if world > 1:
        dist.barrier()
    elapsed_many = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed_many], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed_many = float(t.item())
    for c_ in many_ctx:
        c_.close()
    many_rate = iters * K * R * world / elapsed_many
    nn_rec = {"metric": "search iterations/s (tree search, batched network leaf evaluations, exact Nash at the root)",
              "value": many_rate, "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": 1,
              "ms_per_step": elapsed_many / K * 1e3, "higher_is_better": True, "scaling": "weak", "dtype": "f32", "data": "synthetic",
              "config": {"workload": "configs[4]: MCTS::Search::run with NN::Battle::Network leaves (768-256-256-256-1, seeded synthetic "
                                     "weights), joint UCB (c = 2), 2^18 iterations per search in batches of 16384 descents, exact Nash "
                                     "of the root's empirical matrix on the host; %d random OU turn-1 roots per GPU searched AT ONCE "
                                     "(oakgpu_search_many: one tree, one context, %d host threads per root); one step = one search of every root" % (R, max(1, cores // R)),
                         "roots_per_gpu": R, "host_cores": cores,
                         "nodes": outs_many[-1]["nodes"], "mean_depth": outs_many[-1]["mean_depth"], "nash_value": outs_many[-1]["nash_value"],
                         "leaf_evals_per_s": many_rate},
              "one_root_at_a_time": {"value": single_rate, "unit": "iterations/s", "ms_per_search": elapsed_nn / K * 1e3,
                                     "nodes": outs_nn[-1]["nodes"], "mean_depth": outs_nn[-1]["mean_depth"], "nash_value": outs_nn[-1]["nash_value"],
                                     "note": "the same search, one root per GPU on all host threads (rounds 2-3's figure)"}}
    net.close()
    ctx.close()
    if rank != 0:
        return None
    nn_rec["vs_baseline"] = None
    nn_rec["search_mc_leaves"] = {
        "metric": "search iterations/s (tree search, batched Monte-Carlo leaves)", "value": iters * K * world / elapsed,
        "unit": "iterations/s", "n_gpus": world, "steps": K, "warmup": 1, "ms_per_step": elapsed / K * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u16", "data": "synthetic",
        "config": {"workload": "SURVEY 8(f) rank 1: MCTS::Search::run, Node heap, joint UCB (c = 2), 2^18 iterations per search in "
                               "batches of 16384 descents, rollout leaves, roll clamping {3, 1}; one random OU turn-1 root per GPU",
                   "nodes": outs[-1]["nodes"], "mean_depth": outs[-1]["mean_depth"], "nash_value": outs[-1]["nash_value"]},
    }
    return nn_rec


PROFILE_SOURCE = "profiles/traffic.json"


def profile_json():
    """Per-unit figures derived from the committed rocprofv3 --pmc passes (tools/summarize_profile_r03.py).  Everything the
    bench line takes from here is labelled with a *_source field: it is profile-derived, not measured in the run."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        return {}


def valu_issue_peak():
    """(wave-instructions/s, where it comes from): the best MIXED integer instruction stream of the issue microbenchmark."""
    try:
        pk = json.load(open(os.path.join(ROOT, "profiles", "r04_valu_issue.json")))["peak"]
        return pk[pk["used_by_bench"]] * 1e9, ("profiles/r04_valu_issue.json: peak.%s (best mixed v_add / v_bfe / v_cndmask / v_cmp ... stream, every CU, "
                                               "4-8 waves per SIMD; homogeneous v_add streams reach %.0f G/s, no mixture does)"
                                               % (pk["used_by_bench"], pk["homogeneous_fast_stream_g_per_s"]))
    except Exception:
        return 1024 * 2.4e9 / 4, "nominal: 1,024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (profiles/r04_valu_issue.json not found)"


FP32_MATRIX_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MATRIX_TFLOPS = 2500.0     # v_mfma_f32_32x32x16_bf16, dense


def priced(kernels):
    """One roofline fraction for a step made of several kernels whose layers run on DIFFERENT pipes.  kernels = [{name, us,
    parts: [(what, pipe, work, peak)]}] with work and peak in the same unit per second (FLOP, FLOP/s -- or bytes, bytes/s).
    A kernel's t_min = sum over its parts of work / peak (the time it would take with every part at the peak of the pipe it
    runs on); frac = t_min / measured time, per kernel and -- time-weighted -- for the step: sum of t_min / sum of time.  Work
    is ALGORITHMIC (an fp32 layer computed as bf16 triples counts its six bf16 partial products: that is the algorithm on
    that pipe), never more than what executes, so no fraction can exceed 1."""
    rows, tmin_all, t_all = [], 0.0, 0.0
    for k in kernels:
        t = k["us"] * 1e-6
        tmin = sum(w / pk for _, _, w, pk in k["parts"])
        rows.append({"kernel": k["name"], "us": k["us"], "frac": tmin / t,
                     "parts": [{"what": what, "pipe": pipe, "work": w, "peak_per_s": pk, "us_at_peak": w / pk * 1e6} for what, pipe, w, pk in k["parts"]]})
        tmin_all += tmin
        t_all += t
    return tmin_all / t_all, rows, tmin_all, t_all


def host_threads():
    """(threads, source) for the CPU baselines: the cores this process may actually use = min(affinity mask, cgroup CPU quota)
    (a one-GPU job on the GPU box sees every CPU of the host in its mask but is given a 16-CPU share).  source says where the
    number came from: "env" (BENCH_CPU_THREADS), "cgroup", "affinity", or "assumed" (no quota visible on a > 64-CPU host: the
    documented 16-CPU share of a one-GPU job is assumed rather than oversubscribing a shared host)."""
    if os.environ.get("BENCH_CPU_THREADS"):
        return int(os.environ["BENCH_CPU_THREADS"]), "env"
    try:
        cores = max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        cores = os.cpu_count() or 1
    source = "affinity"
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                q = max(1, int(round(float(quota) / period)))
                if q < cores:
                    cores, source = q, "cgroup"
            break
        except (OSError, ValueError, IndexError):
            continue
    if cores > 64:
        cores, source = 16, "assumed"
    return cores, source


def cpu_baseline(n):
    """The CPU oracle (kind "port": this repo's restatement, NOT the Oak binary) on the host cores,
    same lane seeds as the GPU batch, bounded to roughly 10-20 thread-seconds of work."""
    import oracle_lib as O
    cores, tsrc = host_threads()
    sample = min(n, 65536)
    b, d, p, r = O.make_random_ou_batch(sample, SEED0)
    reps, total = 0, 0
    t0 = time.perf_counter()
    while reps < 40 and (reps < 3 or time.perf_counter() - t0 < 1.5):
        bb, dd = b.copy(), d.copy()
        _, steps = O.rollout_batch(bb, dd, r, p, max_steps=MAX_STEPS, threads=cores)  # p advances in place
        total += int(steps.sum())
        reps += 1
    dt = time.perf_counter() - t0
    # one thread, like the reference's `benchmark` binary (benchmark.cc:9-51: one search thread; TUTORIAL.md:18-21 is its number)
    s1 = min(sample, 4096)
    t1 = time.perf_counter()
    _, st1 = O.rollout_batch(b[:s1].copy(), d[:s1].copy(), r[:s1], p[:s1].copy(), max_steps=MAX_STEPS, threads=1)
    dt1 = time.perf_counter() - t1
    return {
        "value": total / dt,
        "unit": "turn-steps/s",
        "cores": cores,
        "threads_source": tsrc,
        "single_thread": {"value": int(st1.sum()) / dt1, "unit": "turn-steps/s", "sample": "%d playouts, 1 thread" % s1},
        "kind": "port",
        "sample": "%d passes over %d playouts (same lane seeds as the GPU batch), %d threads, oracle/liboracle.so "
                  "gcc -O3 -march=x86-64-v3" % (reps, sample, cores),
    }


def cpu_baseline_leaf(net_path, mid, dur_mid):
    """leaf-evals/s of oracle/nn_host.c (plain-C fp32 port of value_inference: sparse first layers, batch-1 GEMVs like
    the reference's Eigen path, no embedding cache -- every battle is evaluated once) on the same mid-game states."""
    import oracle_lib as O
    cores, tsrc = host_threads()
    sample = min(mid.shape[0], 32768)
    b, d = mid[:sample].cpu().numpy(), dur_mid[:sample].cpu().numpy()
    net = O.CNet(net_path)
    reps = 0
    t0 = time.perf_counter()
    while reps < 20 and (reps < 1 or time.perf_counter() - t0 < 2.0):
        net.value_inference_batch(b, d, threads=cores)
        reps += 1
    dt = time.perf_counter() - t0
    net.close()
    return {"value": sample * reps / dt, "unit": "leaf-evals/s", "cores": cores, "threads_source": tsrc, "kind": "port",
            "sample": "%d passes over %d of the GPU batch's mid-game states, %d threads, oracle/nn_host.c gcc -O3 -march=x86-64-v3"
                      % (reps, sample, cores)}


def cpu_baseline_config3(net_path, n):
    """configs[2] on the host: one random turn-step of every lane (oracle engine) + value_inference of every lane (C port),
    episode of 10 turns from the turn-0 batch."""
    import oracle_lib as O
    cores, tsrc = host_threads()
    sample = min(n, 16384)
    b, d, p, r = O.make_random_ou_batch(sample, SEED0)
    net = O.CNet(net_path)
    res = r.copy()
    turns, total = 0, 0
    t0 = time.perf_counter()
    while turns < 10 and (turns < 2 or time.perf_counter() - t0 < 3.0):
        res, steps = O.rollout_batch(b, d, res, p, max_steps=1, threads=cores)
        net.value_inference_batch(b, d, threads=cores)
        total += int(steps.sum())
        turns += 1
    dt = time.perf_counter() - t0
    net.close()
    return {"value": total / dt, "unit": "turn-steps/s", "cores": cores, "threads_source": tsrc, "kind": "port",
            "sample": "%d turns of %d lanes (turn-step by oracle/liboracle.so + value_inference by oracle/nn_host.c), %d threads"
                      % (turns, sample, cores)}


def cpu_baseline_config4():
    """configs[3] on the host: the oracle's playout loop WITH root prep (battle.rng from the lane's stream +
    randomize_hidden_variables) over 16 of the 256 roots x 4096 replicas, all host threads."""
    import numpy as np
    import oracle_lib as O
    cores, tsrc = host_threads()
    roots, reps = 16, 4096
    b, d, _, r = O.make_random_ou_batch(roots, SEED0)
    _, _, p, _ = O.make_random_ou_batch(roots * reps, 0xC40000000000)
    B, D, R = np.repeat(b, reps, axis=0), np.repeat(d, reps, axis=0), np.repeat(r, reps)
    t0 = time.perf_counter()
    _, steps = O.rollout_batch(B, D, R, p, max_steps=MAX_STEPS, prep=True, threads=cores)
    dt = time.perf_counter() - t0
    return {"value": int(steps.sum()) / dt, "unit": "turn-steps/s", "cores": cores, "threads_source": tsrc, "kind": "port",
            "sample": "%d of the 256 roots x %d playouts with root prep, %d threads, oracle/liboracle.so" % (roots, reps, cores)}


if __name__ == "__main__":
    main()
