"""Host-side mirror of the reference's search-node surface for the rollout / update path.

Names follow the reference (pyoak: `update`, `parse_battle`; libpkmn: `choices`; MCTS:
`rollout`) but every call is batched: arrays carry one battle per row.  All compute goes
through the C ABI of liboakgpu.so (include/oakgpu.h); nothing here computes on the CPU.

  Context.rollout(...)   <- MCTS::Search::init_stats_and_rollout   search/mcts.h:448-496
  Context.update(...)    <- pkmn_gen1_battle_update / PKMN::update libpkmn/pkmn.h:106-139
  Context.choices(...)   <- pkmn_gen1_battle_choices               libpkmn/pkmn.h:141-156
  Context.battle(...)    <- PKMN::battle(p1, p2, seed)             libpkmn/pkmn.h:50-57
  Network.value_inference(...) <- NN::Battle::NetworkImpl::value_inference  nn/battle/network.h:72-79
"""
import ctypes as C
import os

import numpy as np

from . import _lib, gamedata


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _u8(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a if shape is None else a.reshape(shape)


class Context:
    """One per process / GPU.  Host-array API (copies over PCIe); the device-pointer API used by
    bench.py is exposed through `lib` + `handle` with torch tensors' data_ptr()."""

    def __init__(self, device=0):
        self.lib = _lib.load()
        h = C.c_void_p()
        _lib.check(self.lib.oakgpu_create(C.byref(h), device))
        self.handle = h
        self._pools = False

    def close(self):
        if self.handle:
            self.lib.oakgpu_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _lib.check(self.lib.oakgpu_set_stream(self.handle, C.c_void_p(stream_ptr)))

    def set_playouts_per_lane(self, k):
        _lib.check(self.lib.oakgpu_set_playouts_per_lane(self.handle, int(k)))

    def poke_engine_eval(self, battles, root_score=0.0):
        """PokeEngine::Eval on a batch (poke-engine-evaluate.h:186-202): returns (values, raw scores)."""
        battles = _u8(battles)
        n = battles.shape[0]
        values = np.zeros(n, dtype=np.float32)
        scores = np.zeros(n, dtype=np.float32)
        _lib.check(self.lib.oakgpu_poke_engine_eval(self.handle, _p(battles), n, C.c_float(root_score), _p(values), _p(scores)))
        return values, scores

    def set_regroup(self, rounds=1, suspend_below=0, shrink=1):
        """Regrouping rounds of the queue schedule (include/oakgpu.h: oakgpu_set_regroup; default: off, one dispatch);
        results never change."""
        _lib.check(self.lib.oakgpu_set_regroup(self.handle, int(rounds), int(suspend_below), int(shrink)))

    def set_migration(self, mode=1, long_steps=300, adopters=0):
        """Long-playout migration of the queue kernel (include/oakgpu.h: oakgpu_set_migration); results never change."""
        _lib.check(self.lib.oakgpu_set_migration(self.handle, int(mode), int(long_steps), int(adopters)))

    def set_migration_window(self, window=48):
        """Donate a playout whose actives stood still for `window` turn-steps (oakgpu_set_migration_window; 0 = off)."""
        _lib.check(self.lib.oakgpu_set_migration_window(self.handle, int(window)))

    def set_standstill_skip(self, on=True):
        """The queue kernel's exact fast-forward of proven frozen standstills (oakgpu_set_standstill_skip); results never change."""
        _lib.check(self.lib.oakgpu_set_standstill_skip(self.handle, 1 if on else 0))

    def queue_counters(self):
        """The 64 control words of the last queue launch (oakgpu_get_queue_counters): [40] donations, [41] adoptions, [63] sticky error bits."""
        out = np.zeros(64, dtype=np.uint32)
        _lib.check(self.lib.oakgpu_get_queue_counters(self.handle, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_rollout_engine(self, engine=2):
        """2: register-resident engine + queue (default); 1: LDS-resident engine, one launch per batch (the second implementation)."""
        _lib.check(self.lib.oakgpu_set_rollout_engine(self.handle, int(engine)))

    def stream_ptr(self):
        """hipStream_t of this context (wrap with torch.cuda.ExternalStream to share it with torch)."""
        return self.lib.oakgpu_get_stream(self.handle)

    def synchronize(self):
        _lib.check(self.lib.oakgpu_synchronize(self.handle))

    def ensure_ou_pools(self):
        if not self._pools:
            legal, pools, sizes = gamedata.ou_pools()
            _lib.check(self.lib.oakgpu_set_ou_pools(self.handle, _p(legal), len(legal),
                                                    _p(np.ascontiguousarray(pools)), _p(sizes)))
            self._pools = True

    # -- host-array API ------------------------------------------------------------------
    def rollout(self, battles, durations, results, prng, max_steps=1000, prep=False, return_state=False):
        battles = _u8(battles)
        n = battles.shape[0]
        durations = _u8(durations, (n, 8))
        results = _u8(results, (n,))
        prng = _u8(prng, (n, 8)).copy()
        out = np.zeros(n, dtype=np.uint8)
        steps = np.zeros(n, dtype=np.uint32)
        values = np.zeros(n, dtype=np.float32)
        bo = np.zeros((n, 384), dtype=np.uint8) if return_state else None
        do = np.zeros((n, 8), dtype=np.uint8) if return_state else None
        _lib.check(self.lib.oakgpu_rollout(self.handle, _p(battles), _p(durations), _p(results), _p(prng), n,
                                           max_steps, 1 if prep else 0, _p(out), _p(steps), _p(values), _p(bo), _p(do)))
        res = dict(results=out, steps=steps, values=values, prng=prng)
        if return_state:
            res.update(battles=bo, durations=do)
        return res

    def rollout_group(self, batches, max_steps=1000, prep=False, return_state=False):
        """Several independent batches [(battles, durations, results, prng), ...] drained by ONE launch
        (oakgpu_rollout_group); returns one result dict per batch, identical to separate rollout() calls."""
        descs = (_lib.RolloutBatch * len(batches))()
        outs, keep = [], []
        for k, (battles, durations, results, prng) in enumerate(batches):
            battles = _u8(battles)
            n = battles.shape[0]
            durations, results, prng = _u8(durations, (n, 8)), _u8(results, (n,)), _u8(prng, (n, 8)).copy()
            res = dict(results=np.zeros(n, dtype=np.uint8), steps=np.zeros(n, dtype=np.uint32),
                       values=np.zeros(n, dtype=np.float32), prng=prng)
            if return_state:
                res.update(battles=np.zeros((n, 384), dtype=np.uint8), durations=np.zeros((n, 8), dtype=np.uint8))
            keep.append((battles, durations, results))
            a = lambda x: None if x is None or x.size == 0 else x.ctypes.data
            descs[k] = _lib.RolloutBatch(a(battles), a(durations), a(results), a(prng), n, a(res["results"]), a(res["steps"]),
                                         a(res["values"]), a(res.get("battles")), a(res.get("durations")))
            outs.append(res)
        _lib.check(self.lib.oakgpu_rollout_group(self.handle, descs, len(batches), max_steps, 1 if prep else 0))
        return outs

    def rollout_shared_device(self, battle, durations, result, draws, n, max_steps=1000, prep=True, return_state=False):
        """n playouts from ONE root driven by ONE sequential device generator, in the reference's order
        (benchmark.cc:23-31 + mcts.h:250-263,448-496).  `draws` = the generator's uniform_64() output
        (mt19937_uniform_64 below).  Returns results / steps / values / offsets / consumed (+ final states)."""
        draws = np.ascontiguousarray(draws, dtype=np.uint64)
        res = dict(results=np.zeros(n, dtype=np.uint8), steps=np.zeros(n, dtype=np.uint32), values=np.zeros(n, dtype=np.float32),
                   offsets=np.zeros(n, dtype=np.uint32))
        if return_state:
            res.update(battles=np.zeros((n, 384), dtype=np.uint8), durations=np.zeros((n, 8), dtype=np.uint8))
        consumed = C.c_uint64(0)
        _lib.check(self.lib.oakgpu_rollout_shared_device(
            self.handle, _p(_u8(battle, (384,))), _p(_u8(durations, (8,))), int(result), _p(draws), len(draws), n, max_steps,
            1 if prep else 0, _p(res["results"]), _p(res["steps"]), _p(res["values"]), _p(res.get("battles")),
            _p(res.get("durations")), _p(res["offsets"]), C.cast(C.byref(consumed), C.c_void_p)))
        res["consumed"] = consumed.value
        return res

    def update(self, battles, c1, c2, durations, overrides=None, want_actions=True):
        """In-place batched update; returns (results, actions or None)."""
        n = battles.shape[0]
        assert battles.dtype == np.uint8 and battles.flags.c_contiguous
        assert durations.dtype == np.uint8 and durations.flags.c_contiguous
        c1 = _u8(c1, (n,))
        c2 = _u8(c2, (n,))
        res = np.zeros(n, dtype=np.uint8)
        actions = np.zeros((n, 16), dtype=np.uint8) if want_actions else None
        ov = None if overrides is None else _u8(overrides, (n, 16))
        _lib.check(self.lib.oakgpu_update(self.handle, _p(battles), _p(c1), _p(c2), _p(durations), _p(actions),
                                          _p(ov), n, _p(res)))
        return res, actions

    def choices(self, battles, results, player):
        battles = _u8(battles)
        n = battles.shape[0]
        out = np.zeros((n, 9), dtype=np.uint8)
        counts = np.zeros(n, dtype=np.uint8)
        _lib.check(self.lib.oakgpu_choices(self.handle, _p(battles), _p(_u8(results, (n,))), player, _p(out), _p(counts), n))
        return out, counts

    def battle(self, teams, seeds, first_update=True):
        """teams uint8[n, 2, 6, 5]; seeds uint64[n] -> (battles, durations, results)."""
        teams = _u8(teams)
        n = teams.shape[0]
        teams = teams.reshape(n, 60)
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        b = np.zeros((n, 384), dtype=np.uint8)
        d = np.zeros((n, 8), dtype=np.uint8)
        r = np.zeros(n, dtype=np.uint8)
        _lib.check(self.lib.oakgpu_init_battles(self.handle, _p(teams), _p(seeds), n, 1 if first_update else 0,
                                                _p(b), _p(d), _p(r)))
        return b, d, r


def mt19937_uniform_64(seed, count, skip=0):
    """std::mt19937{seed} -> `count` uniform_64() values after skipping `skip` (util/random.h:10-65), host side."""
    out = np.zeros(count, dtype=np.uint64)
    _lib.check(_lib.load().oakgpu_mt19937_fill(int(seed), int(skip), _p(out), count))
    return out


class Network:
    """Battle network on the GPU.  Mirrors NN::Battle::Network (nn/battle/network.h:22-176): constructed
    from a `.battle.net` parameter file; raises RuntimeError (OakGpuError) on an unreadable or malformed
    file like the reference's loader (search.cc:62-148)."""

    def __init__(self, ctx, path=None, data=None):
        self.ctx = ctx
        h = C.c_void_p()
        if path is not None:
            _lib.check(ctx.lib.oakgpu_net_load(ctx.handle, os.fsencode(path), C.byref(h)))
        else:
            buf = (C.c_ubyte * len(data)).from_buffer_copy(data)
            _lib.check(ctx.lib.oakgpu_net_load_memory(ctx.handle, buf, len(data), C.byref(h)))
        self.handle = h

    def shape(self):
        """(fc0.in, fc0.out, value_fc2.out, p1_policy_fc2.out) -- MainNet::shape(), main-net.h:32-34."""
        v = [C.c_int() for _ in range(4)]
        _lib.check(self.ctx.lib.oakgpu_net_shape(self.handle, *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def main_precision(self):
        """(mode in effect: "fp32" | "split" | "pair", whether the bf16-triple form is allowed for this network): oakgpu_net_main_precision."""
        allowed = C.c_int(0)
        mode = self.ctx.lib.oakgpu_net_main_precision(self.handle, C.byref(allowed))
        if mode < 0:
            raise _lib.OakGpuError("oakgpu_net_main_precision failed")
        return ("fp32", "split", "pair")[mode], bool(allowed.value)

    def set_main_precision(self, mode):
        """"pair" (default: fp32 values as scaled fp16 pairs on the fp16 matrix pipe, fp32 accumulation), "split" (bf16 triples) or
        "fp32" (fp32 MFMA); include/oakgpu.h: oakgpu_net_set_main_precision.  Returns the previous mode."""
        prev = self.ctx.lib.oakgpu_net_set_main_precision(self.handle, {"fp32": 0, "split": 1, "pair": 2}[mode])
        if prev < 0:
            raise _lib.OakGpuError("oakgpu_net_set_main_precision failed")
        return ("fp32", "split", "pair")[prev]

    def value_inference(self, battles, durations, return_embedding=False):
        battles = _u8(battles)
        n = battles.shape[0]
        durations = _u8(durations, (n, 8))
        values = np.zeros(n, dtype=np.float32)
        emb = np.zeros((n, self.shape()[0]), dtype=np.float32) if return_embedding else None
        _lib.check(self.ctx.lib.oakgpu_leaf_eval(self.ctx.handle, self.handle, _p(battles), _p(durations), n,
                                                 _p(values), _p(emb)))
        return (values, emb) if return_embedding else values

    def value_policy_inference(self, battles, durations, p1_choices, p1_counts, p2_choices, p2_counts):
        """(values[n], p1_logits[n, 9], p2_logits[n, 9]) -- NetworkImpl::value_policy_inference, network.h:102-123.
        choices / counts as returned by Context.choices()."""
        battles = _u8(battles)
        n = battles.shape[0]
        values = np.zeros(n, dtype=np.float32)
        l1 = np.zeros((n, 9), dtype=np.float32)
        l2 = np.zeros((n, 9), dtype=np.float32)
        _lib.check(self.ctx.lib.oakgpu_leaf_eval_policy(self.ctx.handle, self.handle, _p(battles), _p(_u8(durations, (n, 8))), n,
                                                        _p(_u8(p1_choices, (n, 9))), _p(_u8(p1_counts, (n,))),
                                                        _p(_u8(p2_choices, (n, 9))), _p(_u8(p2_counts, (n,))),
                                                        _p(values), _p(l1), _p(l2)))
        return values, l1, l2

    def close(self):
        if self.handle:
            self.ctx.lib.oakgpu_net_free(self.ctx.handle, self.handle)
            self.handle = None
