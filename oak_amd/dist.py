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
