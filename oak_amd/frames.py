"""`.battle.data` training frames (host side): reader / writer over the C ABI and the GPU self-play game loop.

Mirrors Train::Battle::CompressedFrames (cpp/include/train/battle/compressed-frame.h:37-243) and the per-game loop of
the reference's data generator (cpp/src/generate.cc:238-322); the (de)serialisation and the loop themselves are C++
(oak_amd/csrc/selfplay.hip) -- this module only marshals arrays."""
import ctypes as C

import numpy as np

from . import _lib


def read_frames(data):
    """Parse a `.battle.data` byte string (a concatenation of game records).  Returns a list of games:
    {"battle": uint8[384], "result": int, "updates": [{"m", "n", "c1", "c2", "iterations", "empirical_value",
    "nash_value", "p1_empirical", "p1_nash", "p2_empirical", "p2_nash"}, ...]}."""
    lib = _lib.load()
    buf = np.frombuffer(bytes(data), dtype=np.uint8)
    games, pos = [], 0
    while pos < buf.size:
        battle = np.zeros(384, dtype=np.uint8)
        result, count, used = C.c_uint8(0), C.c_uint32(0), C.c_size_t(0)
        p = buf[pos:].ctypes.data_as(C.c_void_p)
        _lib.check(lib.oakgpu_frames_read(p, buf.size - pos, None, None, None, 0, C.byref(count), C.byref(used)))   # count first
        ups = (_lib.FrameUpdate * max(count.value, 1))()
        _lib.check(lib.oakgpu_frames_read(p, buf.size - pos, battle.ctypes.data_as(C.c_void_p), C.byref(result), ups, count.value,
                                          C.byref(count), C.byref(used)))
        games.append({"battle": battle, "result": int(result.value), "updates": [
            {"m": u.m, "n": u.n, "c1": u.c1, "c2": u.c2, "iterations": u.iterations, "empirical_value": u.empirical_value,
             "nash_value": u.nash_value, "p1_empirical": np.array(u.p1_empirical[:u.m]), "p1_nash": np.array(u.p1_nash[:u.m]),
             "p2_empirical": np.array(u.p2_empirical[:u.n]), "p2_nash": np.array(u.p2_nash[:u.n])} for u in ups[:count.value]]})
        pos += used.value
    return games


def write_frames(battle, result, updates):
    """One game record (bytes) from the first battle (after the opening update), the final result byte and a list of
    update dicts shaped like read_frames' (probability arrays of length m / n)."""
    lib = _lib.load()
    ups = (_lib.FrameUpdate * max(len(updates), 1))()
    for k, u in enumerate(updates):
        ups[k].m, ups[k].n, ups[k].c1, ups[k].c2 = int(u["m"]), int(u["n"]), int(u["c1"]), int(u["c2"])
        ups[k].iterations = int(u["iterations"])
        ups[k].empirical_value, ups[k].nash_value = float(u["empirical_value"]), float(u["nash_value"])
        for name in ("p1_empirical", "p1_nash", "p2_empirical", "p2_nash"):
            arr = getattr(ups[k], name)
            for i, x in enumerate(u[name]):
                arr[i] = float(x)
    size = lib.oakgpu_frames_size(ups, len(updates))
    out = np.zeros(size, dtype=np.uint8)
    written = C.c_size_t(0)
    b = np.ascontiguousarray(battle, dtype=np.uint8).reshape(384)
    _lib.check(lib.oakgpu_frames_write(b.ctypes.data_as(C.c_void_p), int(result), ups, len(updates), out.ctypes.data_as(C.c_void_p), size,
                                       C.byref(written)))
    return out[:written.value].tobytes()


def selfplay_game(ctx, teams, battle_seed, iterations=1 << 12, batch=1024, bandit="ucb", c=2.0, evaluator="mc", policy_mode="e",
                  policy_temp=1.0, policy_min=0.0, max_battle_length=0, seed=1, alpha=0.05, root_rolls=3, other_rolls=1, keep_node=False,
                  stats=None):
    """One self-play game on the GPU path (oakgpu_selfplay_game).  teams: uint8[2, 6, 5] (species + 4 moves per set).
    keep_node: generate's --keep-node (one heap for the game, Heap::update after every turn); stats: optional dict that
    receives "nodes_kept".  Returns (record bytes, number of frames, final result byte)."""
    use_net = not isinstance(evaluator, str)
    prm = _lib.SelfplayParams()
    prm.search = _lib.SearchParams(iterations=int(iterations), batch=int(batch), ucb_c=float(c),
                                   bandit={"ucb": 0, "pucb": 1, "ucb1": 2, "exp3": 3, "pexp3": 4}[bandit],
                                   eval=1 if use_net else {"mc": 0, "poke-engine": 2}[evaluator], max_depth=0, root_rolls=int(root_rolls),
                                   other_rolls=int(other_rolls), seed=0, matrix_ucb=0, mucb_delay=0, mucb_minimum=0, mucb_c=0.0,
                                   exp3_alpha=float(alpha))
    prm.policy_mode = policy_mode.encode()
    prm.policy_temp, prm.policy_min = float(policy_temp), float(policy_min)
    prm.max_battle_length, prm.seed = int(max_battle_length), int(seed)
    prm.keep_node = 1 if keep_node else 0
    t = np.ascontiguousarray(teams, dtype=np.uint8).reshape(60)
    cap = 4 + 2 + 384 + 1 + 83 * (int(max_battle_length) or 2048)
    out = np.zeros(cap, dtype=np.uint8)
    written, frames, result = C.c_size_t(0), C.c_uint32(0), C.c_uint8(0)
    _lib.check(ctx.lib.oakgpu_selfplay_game(ctx.handle, evaluator.handle if use_net else None, t.ctypes.data_as(C.c_void_p), int(battle_seed),
                                            C.byref(prm), out.ctypes.data_as(C.c_void_p), cap, C.byref(written), C.byref(frames), C.byref(result)))
    if stats is not None:
        stats["nodes_kept"] = int(prm.nodes_kept)
    return out[:written.value].tobytes(), int(frames.value), int(result.value)


def selfplay_games(ctxs, teams, battle_seeds, seeds, iterations=1 << 12, batch=1024, bandit="ucb", c=2.0, evaluator="mc", policy_mode="e",
                   policy_temp=1.0, policy_min=0.0, max_battle_length=0, alpha=0.05, root_rolls=3, other_rolls=1, keep_node=False,
                   threads_per_game=0):
    """n self-play games at once on one GPU (oakgpu_selfplay_games): game g on ctxs[g] with teams[g] (uint8[n, 2, 6, 5]), battle_seeds[g]
    and policy seed seeds[g].  Returns a list of (record bytes, number of frames, final result byte) -- each what selfplay_game(ctxs[g],
    teams[g], battle_seeds[g], seed=seeds[g], ...) returns alone."""
    n = len(ctxs)
    use_net = not isinstance(evaluator, str)
    prms = (_lib.SelfplayParams * n)()
    for g in range(n):
        prms[g].search = _lib.SearchParams(iterations=int(iterations), batch=int(batch), ucb_c=float(c),
                                           bandit={"ucb": 0, "pucb": 1, "ucb1": 2, "exp3": 3, "pexp3": 4}[bandit],
                                           eval=1 if use_net else {"mc": 0, "poke-engine": 2}[evaluator], max_depth=0, root_rolls=int(root_rolls),
                                           other_rolls=int(other_rolls), seed=0, matrix_ucb=0, mucb_delay=0, mucb_minimum=0, mucb_c=0.0,
                                           exp3_alpha=float(alpha))
        prms[g].policy_mode = policy_mode.encode()
        prms[g].policy_temp, prms[g].policy_min = float(policy_temp), float(policy_min)
        prms[g].max_battle_length, prms[g].seed = int(max_battle_length), int(seeds[g])
        prms[g].keep_node = 1 if keep_node else 0
    t = np.ascontiguousarray(teams, dtype=np.uint8).reshape(n, 60)
    bs = (C.c_uint64 * n)(*[int(x) for x in battle_seeds])
    cap = 4 + 2 + 384 + 1 + 83 * (int(max_battle_length) or 2048)
    out = np.zeros((n, cap), dtype=np.uint8)
    written, frames, result = (C.c_size_t * n)(), (C.c_uint32 * n)(), (C.c_uint8 * n)()
    cp = (C.c_void_p * n)(*[c_.handle for c_ in ctxs])
    _lib.check(ctxs[0].lib.oakgpu_selfplay_games(cp, evaluator.handle if use_net else None, t.ctypes.data_as(C.c_void_p), bs, prms, n, int(threads_per_game),
                                                 out.ctypes.data_as(C.c_void_p), cap, written, frames, result))
    return [(out[g, :written[g]].tobytes(), int(frames[g]), int(result[g])) for g in range(n)]
