"""Multi-GPU sharding of the playout batch (one process per GPU, torch.distributed over RCCL).

Playouts are independent (the reference runs them as unrelated threads, cpp/src/generate.cc:527-536),
so the lane range [0, n_total) is cut into `world` contiguous blocks -- root-parallel MCTS keeps each
root's playouts on one device.  The only exchange step of the path is ONE all-gather of the
fp32 leaf values back to every root; it goes through torch.distributed (backend "nccl" == RCCL over
xGMI on the GPU box, "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous lane block of `rank`: [lo, hi).  Blocks differ by at most one lane."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def lane_seed0(seed0, n_total, rank, world):
    """First lane seed of this rank's block (lane i of the global batch is seeded seed0 + i)."""
    return seed0 + shard_range(n_total, rank, world)[0]


def gather_values(values, n_total=None, force=False):
    """All-gather per-rank fp32 leaf values into the global lane order on every rank."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not force):
        return values
    world = dist.get_world_size()
    if n_total is None or n_total % world == 0:
        out = torch.empty(values.numel() * world, dtype=values.dtype, device=values.device)
        dist.all_gather_into_tensor(out, values.contiguous())
        return out
    # ragged blocks: pad to the largest block, gather, then drop the padding
    sizes = [shard_range(n_total, r, world)[1] - shard_range(n_total, r, world)[0] for r in range(world)]
    m = max(sizes)
    padded = torch.zeros(m, dtype=values.dtype, device=values.device)
    padded[:values.numel()] = values
    out = torch.empty(m * world, dtype=values.dtype, device=values.device)
    dist.all_gather_into_tensor(out, padded)
    return torch.cat([out[r * m:r * m + sizes[r]] for r in range(world)])


def gather_round(stage, out=None, async_op=False):
    """One collective for a whole ROUND of batches in flight: `stage` is [S, n] (row k = the leaf values of batch k of
    this rank), the result is [world, S, n] on every rank (out[r, k] = rank r's batch k).  Returns (out, work); with
    async_op the caller keeps `work` and waits for it later (bench.py never makes a batch stream wait for it)."""
    world = dist.get_world_size()
    S, n = stage.shape
    if out is None:
        out = torch.empty((world, S, n), dtype=stage.dtype, device=stage.device)
    work = dist.all_gather_into_tensor(out.view(-1), stage.contiguous().view(-1), async_op=async_op)
    return out, work


def round_to_global(out):
    """[world, S, n] from gather_round -> [S, world * n]: batch k's values in global lane order (rank-major blocks)."""
    world, S, n = out.shape
    return out.permute(1, 0, 2).reshape(S, world * n)


def per_root_means(all_values, playouts_per_root):
    """Root-parallel MCTS (BASELINE config 4): mean leaf value per root from the gathered lane values."""
    return all_values.view(-1, playouts_per_root).mean(dim=1)


# ---- BASELINE config 4: root-parallel MCTS, roots sharded contiguous-by-root, per-root pre-reduction on the device,
# ONE all-gather of one float per root (256 floats: latency-bound) -------------------------------------------------
def root_shard(n_roots, rank, world):
    """Roots of `rank`: [lo, hi) -- contiguous, so a root's playouts stay on one device and are reduced there."""
    return shard_range(n_roots, rank, world)


def gather_root_means(local_means, n_roots):
    """All-gather the per-root mean leaf values: `local_means` holds this rank's roots (root_shard order); returns the
    n_roots means in global root order on every rank.  One collective of at most ceil(n_roots / world) floats per rank."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_means
    return gather_values(local_means, n_roots)


def group_padding(n_roots, world, groups):
    """Floats per rank in one group's all-gather: the largest group of the rank with the most roots."""
    most = -(-n_roots // world)
    return -(-most // max(1, min(groups, most)))


def assemble_group_means(n_roots, world, groups, blocks):
    """blocks[g] = one gathered block [world * per] of group g (RootGroups history entry) -> the n_roots means in global root
    order: rank r's roots are root_shard(n_roots, r, world), its group g the g-th contiguous part of them."""
    import numpy as np
    out = np.full(n_roots, np.nan, dtype=np.float32)
    per = len(blocks[0]) // world
    for r in range(world):
        lo, hi = root_shard(n_roots, r, world)
        G = max(1, min(groups, hi - lo)) if hi > lo else 1
        for g in range(G):
            a, b = shard_range(hi - lo, g, G)
            out[lo + a:lo + b] = blocks[g][r * per:r * per + (b - a)]
    return out


class RootGroups:
    """BASELINE configs[3] without a barrier across roots.  The roots of root-parallel MCTS are independent trees (the
    reference's workers never wait for each other, cpp/src/generate.cc:527-536): nothing requires every root to finish search
    step k before any root starts step k + 1.  This rank's roots are cut into `groups` contiguous GROUPS, each with its own
    context (HIP stream); a group's step = root prep + playouts of its roots (mcts.h:250-263, 448-496) -> one mean per root on
    the device -> the group's exchange (one all-gather of its means) -> means on the host -> event; the group's NEXT step is
    launched as soon as that event has completed, whatever the other groups are doing, so one group's bulk fills the SIMDs that
    another group's tail (its longest playouts) leaves idle.  Per-root results do not depend on the grouping: a playout's
    streams are its own (seeded by global lane index) and a root's mean is reduced in a fixed order.

    battles / durations / results_in / prng: device tensors over ALL this rank's lanes (roots x reps, root-major); a group
    works on its contiguous slice in place (prng advances from step to step).  exchange(means, out) -- nullable -- gathers a
    group's padded means [per] into out [world * per] on the current stream; ordered=True launches in a fixed (step, group)
    order so that every rank issues its collectives in the same order (required whenever exchange is a collective).

    Stream ordering: the groups' streams are the contexts' own NON-BLOCKING streams, unordered with torch's current stream.  The
    inputs (and the zero fills of this object's own buffers) are produced on the caller's current stream, so the constructor and
    every run() first make each group stream wait for it (`wait_stream`); inputs must be complete ON THE CALLER'S CURRENT STREAM
    when run() is called."""

    def __init__(self, make_context, device, battles, durations, results_in, prng, roots, reps, groups, world=1, exchange=None,
                 max_steps=1000, per=None, owns_contexts=True):
        import ctypes as C
        self.C, self.torch, self.dev = C, torch, device
        self.reps, self.world, self.exchange, self.max_steps, self.owns = reps, world, exchange, max_steps, owns_contexts
        G = max(1, min(int(groups), roots)) if roots else 1
        self.bounds = [shard_range(roots, g, G) for g in range(G)]
        # padded group size: the same on every rank when the exchange is a collective (group_padding), else this rank's largest group
        self.per = int(per) if per else max((hi - lo for lo, hi in self.bounds), default=0)
        n = roots * reps
        self.results = torch.empty((n,), dtype=torch.uint8, device=device)
        self.steps_out = torch.zeros((n,), dtype=torch.int32, device=device)
        self.values = torch.empty((n,), dtype=torch.float32, device=device)
        self.total = torch.zeros((G,), dtype=torch.int64, device=device)
        self.groups = []
        for g, (lo, hi) in enumerate(self.bounds):
            ctx = make_context()
            st = torch.cuda.ExternalStream(ctx.stream_ptr(), device=device)
            a, b = lo * reps, hi * reps
            self.groups.append(dict(
                ctx=ctx, stream=st, lo=lo, hi=hi, n=b - a, battles=battles[a:b], durations=durations[a:b], rin=results_in[a:b],
                prng=prng[a:b], results=self.results[a:b], steps=self.steps_out[a:b], values=self.values[a:b],
                means=torch.zeros((max(self.per, 1),), dtype=torch.float32, device=device),
                allm=torch.empty((world * max(self.per, 1),), dtype=torch.float32, device=device),
                host=torch.empty((world * max(self.per, 1),), dtype=torch.float32).pin_memory(),
                event=torch.cuda.Event(), done=0, inflight=False, history=[]))
        self._order_after_caller()

    def _order_after_caller(self):
        """Every group stream waits for what the caller's current stream has queued so far (input tensors, zero fills)."""
        cur = self.torch.cuda.current_stream(self.dev)
        for G in self.groups:
            G["stream"].wait_stream(cur)

    def _p(self, t):
        return self.C.c_void_p(t.data_ptr())

    def launch(self, g):
        from . import _lib
        G, P = self.groups[g], self._p
        lib, h = G["ctx"].lib, G["ctx"].handle
        if G["n"]:
            _lib.check(lib.oakgpu_rollout_dev(h, P(G["battles"]), P(G["durations"]), P(G["rin"]), P(G["prng"]), G["n"], self.max_steps, 1,
                                              P(G["results"]), P(G["steps"]), P(G["values"]), None, None))
            _lib.check(lib.oakgpu_segment_mean_dev(h, P(G["values"]), G["hi"] - G["lo"], self.reps, P(G["means"])))
        with self.torch.cuda.stream(G["stream"]):
            if G["n"]:
                self.total[g] += G["steps"].sum(dtype=self.torch.int64)
            if self.exchange is not None:
                self.exchange(G["means"], G["allm"])
            else:
                G["allm"].copy_(G["means"])
            G["host"].copy_(G["allm"], non_blocking=True)
            G["event"].record(G["stream"])
        G["inflight"] = True

    def _finish(self, g, keep):
        G = self.groups[g]
        G["event"].synchronize()                       # the host now holds this group's means of every rank: its step is over
        if keep:
            G["history"].append(G["host"].numpy().copy())
        G["done"] += 1
        G["inflight"] = False

    def run(self, steps, ordered=False, keep=False):
        """Every group performs `steps` search steps.  ordered: fixed (step, group) launch order (collectives match across
        ranks); else work-conserving: whichever group's means have arrived is relaunched first."""
        for G in self.groups:
            G["done"], G["inflight"], G["history"] = 0, False, []
        self._order_after_caller()
        if ordered:
            for k in range(steps):
                for g in range(len(self.groups)):
                    if self.groups[g]["inflight"]:
                        self._finish(g, keep)
                    self.launch(g)
            for g in range(len(self.groups)):
                if self.groups[g]["inflight"]:
                    self._finish(g, keep)
            return
        launched = [0] * len(self.groups)
        for g in range(len(self.groups)):
            if steps > 0:
                self.launch(g)
                launched[g] = 1
        while any(G["inflight"] for G in self.groups):
            progressed = False
            for g, G in enumerate(self.groups):
                if G["inflight"] and G["event"].query():
                    self._finish(g, keep)
                    if launched[g] < steps:
                        self.launch(g)
                        launched[g] += 1
                    progressed = True
            if not progressed:          # nothing ready: block on the group that was launched first among those in flight
                g = min((g for g, G in enumerate(self.groups) if G["inflight"]), key=lambda q: self.groups[q]["done"])
                self._finish(g, keep)
                if launched[g] < steps:
                    self.launch(g)
                    launched[g] += 1

    def close(self):
        # the groups' streams belong to their contexts: torch must hold nothing that refers to them when they are destroyed (its
        # caching allocator keeps blocks freed on a stream tied to that stream)
        self.torch.cuda.synchronize(self.dev)
        ctxs = [G["ctx"] for G in self.groups]
        self.groups = []
        self.results = self.steps_out = self.values = self.total = None
        self.torch.cuda.empty_cache()
        if self.owns:
            for c in ctxs:
                c.close()


class RootSteps:
    """BASELINE configs[3] as search steps that do not wait for their longest playout (include/oakgpu.h: oakgpu_root_steps_*;
    csrc/oakgpu.hip: k_root_step).  A step hands every root `reps` fresh playouts (root prep + rollout, mcts.h:250-263, 448-496);
    every launch advances each playout in flight by at most `slice` turn-steps; a playout of `len` turn-steps started in step k is
    credited to step k + (len - 1) // slice -- a function of its own length -- and travels between launches on a carry list.  The
    reference's workers never wait for each other either (cpp/src/generate.cc:527-536).  Values never change, only the step they
    are credited to; per-root aggregates are integer (count, sum of 2 x value), so they do not depend on any order.

    root_battles / root_durations / root_results: device tensors of this rank's `roots` roots; lane_prng [roots * reps, 8]: one
    fast_prng stream per (root, replica), advanced by one uniform_64 per step (the seeds are the caller's: by GLOBAL lane index, so
    results do not depend on the number of ranks).  exchange(send, recv) -- nullable -- all-gathers this rank's padded aggregates
    (`per` int64 per rank) on the current stream: the path's ONE collective.  step() is asynchronous; finish() returns the step's
    record once its aggregates are on the host."""

    def __init__(self, ctx, device, root_battles, root_durations, root_results, lane_prng, roots, reps, slice=16, max_steps=1000,
                 world=1, exchange=None, per=None):
        import ctypes as C
        from . import _lib
        self.C, self.lib, self.ctx, self.dev = C, ctx.lib, ctx, device
        self.roots, self.reps, self.slice, self.world, self.exchange = roots, reps, slice, world, exchange
        self.per = int(per) if per else roots
        self.handle = C.c_void_p()
        _lib.check(self.lib.oakgpu_root_steps_create(ctx.handle, roots, reps, slice, max_steps, C.byref(self.handle)))
        self.rb, self.rd, self.rr, self.prng = root_battles, root_durations, root_results, lane_prng
        self.stream = torch.cuda.ExternalStream(ctx.stream_ptr(), device=device)
        n = max(self.per, roots) + 2
        self.report = torch.zeros((n,), dtype=torch.int64, device=device)       # [r] count | sum2 << 32, [roots] turn-steps, [roots + 1] carried | err << 32
        self.send = torch.zeros((self.per,), dtype=torch.int64, device=device)
        self.recv = torch.zeros((world * self.per,), dtype=torch.int64, device=device)
        self.host = torch.zeros((world * self.per + 2,), dtype=torch.int64).pin_memory()
        self.event = torch.cuda.Event()
        self.stream.wait_stream(torch.cuda.current_stream(device))               # inputs and zero fills come from the caller's stream
        self.turn_steps = 0
        self.inflight = False

    def _p(self, t):
        return self.C.c_void_p(t.data_ptr()) if t is not None else None

    def step(self, fresh=True):
        """Launch one search step (fresh=False: a drain step -- no new playouts, the carried ones advance one more slice)."""
        from . import _lib
        assert not self.inflight, "finish() the previous step first (its record lives in one pinned buffer)"
        _lib.check(self.lib.oakgpu_root_steps_launch_dev(self.handle, self._p(self.rb), self._p(self.rd), self._p(self.rr), self._p(self.prng),
                                                         1 if fresh else 0, self._p(self.report)))
        with torch.cuda.stream(self.stream):
            self.send.zero_()
            self.send[:self.roots] = self.report[:self.roots]
            if self.exchange is not None:
                self.exchange(self.send, self.recv)
            else:
                self.recv[:self.per] = self.send
            self.host[:self.world * self.per].copy_(self.recv, non_blocking=True)
            self.host[self.world * self.per:].copy_(self.report[self.roots:self.roots + 2], non_blocking=True)
            self.event.record(self.stream)
        self.inflight = True

    def finish(self):
        """Wait for the step's aggregates: {"count": [world * per], "sum2": [world * per] (rank-major, padded), "turn_steps", "carried"}."""
        import numpy as np
        self.event.synchronize()
        self.inflight = False
        h = self.host.numpy()
        k = self.world * self.per
        acc = h[:k].view(np.uint64)
        tail = h[k:].view(np.uint64)
        err = int(tail[1] >> np.uint64(32))
        if err:
            raise RuntimeError("oakgpu_root_steps: the carry list overflowed (error word %d): playouts were lost" % err)
        self.turn_steps += int(tail[0])
        carried = int(tail[1] & np.uint64(0xFFFFFFFF))
        cap = self.C.c_uint32()
        self.lib.oakgpu_root_steps_capacity(self.handle, self.C.byref(cap))
        need = 2 * (carried + self.roots * self.reps)      # the next launch can carry what is in flight + a whole step's worth; x 2 for the shards' imbalance
        if need > cap.value:             # stalemate-heavy roots: make room before the next launch (keeps the playouts in flight)
            from . import _lib
            _lib.check(self.lib.oakgpu_root_steps_reserve(self.handle, 2 * need))
        return {"count": (acc & np.uint64(0xFFFFFFFF)).astype(np.int64), "sum2": (acc >> np.uint64(32)).astype(np.int64),
                "turn_steps": int(tail[0]), "carried": int(tail[1] & np.uint64(0xFFFFFFFF))}

    def close(self):
        torch.cuda.synchronize(self.dev)
        if self.handle:
            self.lib.oakgpu_root_steps_destroy(self.handle)
            self.handle = None
        # the stream belongs to the context: torch must hold nothing that refers to it when the context is destroyed (the caching
        # allocators -- device and pinned host -- keep blocks used on a stream tied to that stream)
        self.report = self.send = self.recv = self.host = self.event = self.stream = None
        torch.cuda.empty_cache()
        if hasattr(torch._C, "_host_emptyCache"):
            torch._C._host_emptyCache()


def assemble_rank_blocks(n_roots, world, per, flat):
    """flat = a gathered [world * per] array (rank-major, each rank's roots first, then padding) -> the n_roots entries in global
    root order: rank r holds root_shard(n_roots, r, world)."""
    import numpy as np
    flat = np.asarray(flat)
    out = np.zeros(n_roots, dtype=flat.dtype)
    for r in range(world):
        lo, hi = root_shard(n_roots, r, world)
        out[lo:hi] = flat[r * per:r * per + (hi - lo)]
    return out


def credited_means(count, sum2):
    """Per-root mean leaf value of one step's credited playouts (0.5 where a root was credited nothing)."""
    import numpy as np
    c = np.asarray(count, dtype=np.float64)
    return np.where(c > 0, np.asarray(sum2, dtype=np.float64) / (2.0 * np.maximum(c, 1.0)), 0.5).astype(np.float32)
