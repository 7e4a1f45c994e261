"""GPU parity tests proper: HIP path (through the C ABI) vs the CPU oracle, bit-exact."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from oak_amd.parse import parse_battle, result_from_state

pytestmark = pytest.mark.gpu


def _seed_prng(n, seed0):
    prng = np.zeros((n, 8), dtype=np.uint8)
    O.LIB.oracle_fast_prng_seed_batch(O.ptr(prng), n, C.c_uint64(seed0))
    return prng


def test_rollout_bit_exact_random_ou(gpu_ctx):
    n = 4096
    b, d, p, r = O.make_random_ou_batch(n)
    got = gpu_ctx.rollout(b, d, r, p, max_steps=1000, return_state=True)
    ob, od, op = b.copy(), d.copy(), p.copy()
    oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, threads=8)
    assert (got["steps"] == osteps).all()
    assert (got["results"] == oout).all()
    bad = np.nonzero((got["battles"] != ob).any(axis=1))[0]
    assert bad.size == 0, "first differing lane %d" % bad[0]
    assert (got["durations"] == od).all()
    assert (got["prng"] == op).all()
    t = oout & 15
    exp = np.where(t == 1, 1.0, np.where(t == 2, 0.0, 0.5)).astype(np.float32)
    assert (got["values"] == exp).all()


def test_rollout_ragged_and_capped(gpu_ctx):
    # non-multiple-of-block batch, small step cap (lanes stop mid-playout with type NONE)
    n = 777
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x5EED0000)
    got = gpu_ctx.rollout(b, d, r, p, max_steps=17, return_state=True)
    ob, od, op = b.copy(), d.copy(), p.copy()
    oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=17, threads=4)
    assert (got["steps"] == osteps).all() and (got["results"] == oout).all()
    assert (got["battles"] == ob).all() and (got["durations"] == od).all() and (got["prng"] == op).all()


@pytest.mark.gpu
@pytest.mark.parametrize("n,max_steps,prep", [(1, 1, False), (63, 1, True), (64, 2, False), (65, 3, True), (4099, 1, False), (4099, 16, True)])
def test_short_launches_staged_kernel(gpu_ctx, n, max_steps, prep):
    """Launches capped at <= 16 turn-steps take k_rollout_staged (battles staged through LDS with coalesced accesses,
    BASELINE configs[2]'s turn-by-turn stepping): ragged wave counts, root prep, several turns in a row, against the oracle."""
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x57A6ED00 + n)
    ob, od, op, orr = b.copy(), d.copy(), p.copy(), r.copy()
    for turn in range(3):
        got = gpu_ctx.rollout(b, d, r, p, max_steps=max_steps, prep=prep and turn == 0, return_state=True)
        oout, osteps = O.rollout_batch(ob, od, orr, op, max_steps=max_steps, prep=prep and turn == 0, threads=4)
        assert (got["steps"] == osteps).all() and (got["results"] == oout).all()
        assert (got["battles"] == ob).all() and (got["durations"] == od).all() and (got["prng"] == op).all()
        b, d, p, r, orr = got["battles"], got["durations"], got["prng"], got["results"], oout


def test_rollout_empty(gpu_ctx):
    got = gpu_ctx.rollout(np.zeros((0, 384), np.uint8), np.zeros((0, 8), np.uint8), np.zeros(0, np.uint8),
                          np.zeros((0, 8), np.uint8))
    assert got["steps"].size == 0


def test_stepwise_update_choices_actions(gpu_ctx):
    """Drive 60 turn-steps with batched choices()/update() and compare every intermediate
    battle, durations, actions key and result byte with the oracle."""
    n = 1024
    b, d, p, r = O.make_random_ou_batch(n, seed0=0xABCD0000)
    gb, gd, gr = b.copy(), d.copy(), r.copy()
    opts = [O.Options(d[i]) for i in range(n)]
    rng = np.random.default_rng(5)
    for step in range(60):
        ch1, n1 = gpu_ctx.choices(gb, gr, 0)
        ch2, n2 = gpu_ctx.choices(gb, gr, 1)
        c1 = np.zeros(n, np.uint8)
        c2 = np.zeros(n, np.uint8)
        overrides = np.zeros((n, 16), np.uint8)
        use_over = step % 5 == 0
        for i in range(n):
            o1 = O.choices(b[i], 0, (int(r[i]) >> 4) & 3)
            o2 = O.choices(b[i], 1, (int(r[i]) >> 6) & 3)
            assert n1[i] == len(o1) and (ch1[i, :n1[i]] == o1).all(), (step, i)
            assert n2[i] == len(o2) and (ch2[i, :n2[i]] == o2).all(), (step, i)
            c1[i] = o1[rng.integers(len(o1))]
            c2[i] = o2[rng.integers(len(o2))]
            if use_over:
                overrides[i, 0] = 217 + rng.integers(39)
                overrides[i, 8] = 236
        gres, gact = gpu_ctx.update(gb, c1, c2, gd, overrides=overrides if use_over else None)
        for i in range(n):
            if (int(r[i]) & 15) != 0:
                continue
            opts[i].set(None, overrides[i] if use_over else None)
            r[i] = O.update(b[i], int(c1[i]), int(c2[i]), opts[i])
            d[i] = opts[i].durations
        live = (gr & 15) == 0
        assert (gres[live] == r[live]).all(), step
        assert (gb[live] == b[live]).all(), step
        assert (gd[live] == d[live]).all(), step
        acts = np.stack([o.actions for o in opts])
        assert (gact[live] == acts[live]).all(), step
        # finished lanes: keep them frozen on both sides
        gb[~live] = b[~live]
        gd[~live] = d[~live]
        gr = np.where(live, gres, gr).astype(np.uint8)
        r = gr.copy()


def test_tree_step_levels_match_the_oracle(gpu_ctx):
    """oakgpu_tree_step_dev (one search level: joint action, result, chance-action key, both players' next choices) over random
    walks, finished lanes and every damage-roll clamp, every output byte against the oracle -- the default kernel (round 5: the
    register-resident engine, staged) in this process, the LDS-resident engine's kernel in a child with OAKGPU_TREE_STEP=lds."""
    import subprocess
    import tree_step_check
    assert tree_step_check.run(gpu_ctx) > 20000
    assert tree_step_check.run(gpu_ctx, n=64 * 5 + 1, levels=12, seed0=0x7EE51000) > 1000      # a ragged last wave of one lane
    env = dict(os.environ, OAKGPU_TREE_STEP="lds")
    p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tree_step_check.py"), "700", "25"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "kernel: lds" in p.stdout, (p.stdout[-1000:], p.stderr[-3000:])


def test_init_battle_matches_oracle_and_golden(gpu_ctx):
    from oak_amd import gamedata as G
    rng = np.random.default_rng(11)
    legal, pools, sizes = G.ou_pools()
    n = 300
    teams = np.zeros((n, 2, 6, 5), np.uint8)
    for i in range(n):
        for s in range(2):
            sp = rng.choice(legal, 6, replace=False)
            for k in range(6):
                teams[i, s, k, 0] = sp[k]
                m = min(4, sizes[sp[k]])
                teams[i, s, k, 1:1 + m] = rng.choice(pools[sp[k], :sizes[sp[k]]], m, replace=False)
    seeds = rng.integers(0, 2**63, n).astype(np.uint64)
    b0, d0, r0 = gpu_ctx.battle(teams, seeds, first_update=False)
    b1, d1, r1 = gpu_ctx.battle(teams, seeds, first_update=True)
    for i in range(n):
        ob = O.init_battle(teams[i], int(seeds[i]))
        assert (b0[i] == ob).all(), i
        opt = O.Options()
        rr = O.update(ob, 0, 0, opt)
        assert rr == r1[i] and (b1[i] == ob).all() and (d1[i] == opt.durations).all(), i


def test_known_answer_positions_on_gpu(gpu_ctx):
    """cpp/src/search-test.cc:50-109 -- every position has one legal joint action, so the
    search value equals the mean playout value."""
    tests = [("starmie seismictoss 1hp (conf:5) | snorlax bodyslam 1hp", 1.0, 0.0),
             ("starmie seismictoss 1hp (conf:4) | snorlax bodyslam 1hp", .5 + .5 / 2, .03),
             ("starmie seismictoss 1hp (conf:3) | snorlax bodyslam 1hp", .33 + .66 / 2, .03),
             ("starmie seismictoss 1hp (conf:2) | snorlax bodyslam 1hp", .25 + .75 / 2, .03),
             ("starmie seismictoss 1hp (conf:1) | snorlax bodyslam 1hp", .5, .03),
             ("starmie seismictoss 1hp slp6 | snorlax seismictoss 1hp", 0.0, 0.0)]
    for k in range(7):
        tests.append(("starmie seismictoss 101hp slp%d | snorlax seismictoss 1hp" % k, 1.0 / (7 - k), 0.0 if k == 6 else .03))
    n = 1 << 15
    prng = _seed_prng(n, 4242)
    for pos, expected, err in tests:
        b, d = parse_battle(pos, 99)
        res = result_from_state(b)
        got = gpu_ctx.rollout(np.tile(b, (n, 1)), np.tile(d, (n, 1)), np.full(n, res, np.uint8), prng, prep=True)
        assert abs(float(got["values"].mean()) - expected) <= err + 1e-9, (pos, got["values"].mean())


def test_root_parallel_prep_bit_exact(gpu_ctx):
    """BASELINE config 4 shape at test size: R roots x P replicas, each replica re-seeded and its hidden
    counters resampled on the device (mcts.h:250-263 + durations.h:25-97), then rolled out."""
    roots, reps = 24, 96
    b, d, p, r = O.make_random_ou_batch(roots, seed0=0xC0FFEE00)
    # advance the roots so that sleep / confusion / disable / binding durations are live
    O.rollout_batch(b, d, r, p, max_steps=12, threads=2)
    r = np.array([O.LIB.oracle_result_from_state(O.ptr(b[i])) for i in range(roots)], dtype=np.uint8)
    keep = (r & 15) == 0
    b, d, r = b[keep], d[keep], r[keep]
    n = b.shape[0] * reps
    B, D, R = np.repeat(b, reps, axis=0), np.repeat(d, reps, axis=0), np.repeat(r, reps)
    prng = _seed_prng(n, 0xABCDEF)
    got = gpu_ctx.rollout(B, D, R, prng, max_steps=1000, prep=True, return_state=True)
    ob, od, op = B.copy(), D.copy(), prng.copy()
    oout, osteps = O.rollout_batch(ob, od, R, op, max_steps=1000, prep=True, threads=8)
    assert (got["steps"] == osteps).all() and (got["results"] == oout).all()
    assert (got["battles"] == ob).all() and (got["durations"] == od).all() and (got["prng"] == op).all()
    per_root = got["values"].reshape(-1, reps).mean(axis=1)
    assert ((per_root >= 0) & (per_root <= 1)).all()


def test_queue_refill_matches_plain_launch(gpu_ctx):
    """k_rollout_queue (persistent lanes refilled from an atomic playout queue) == one lane per playout."""
    n = 3000
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x51515151)
    gpu_ctx.set_playouts_per_lane(1)
    plain = gpu_ctx.rollout(b, d, r, p, max_steps=1000, return_state=True)
    for k in (2, 5, 64):
        gpu_ctx.set_playouts_per_lane(k)
        q = gpu_ctx.rollout(b, d, r, p, max_steps=1000, return_state=True)
        for key in ("results", "steps", "values", "battles", "durations", "prng"):
            assert (q[key] == plain[key]).all(), (k, key)
    gpu_ctx.set_playouts_per_lane(2)


def test_regrouping_rounds_do_not_change_results(gpu_ctx):
    """Tail regrouping (suspended playouts parked as battle images, resumed 64 to a wave by later dispatches):
    every schedule gives the byte-identical outputs of the plain one-lane-per-playout launch and of the oracle."""
    n = 6000
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x7E57AB1E)
    ob, od, op = b.copy(), d.copy(), p.copy()
    oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, threads=8)
    gpu_ctx.set_playouts_per_lane(1)
    plain = gpu_ctx.rollout(b, d, r, p, max_steps=1000, return_state=True)
    assert (plain["steps"] == osteps).all() and (plain["battles"] == ob).all()
    try:
        for ppl, rounds, below, shrink in ((2, 1, 0, 1), (2, 2, 64, 2), (2, 4, 32, 3), (3, 8, 63, 1), (4, 3, 17, 5), (2, 4, 1, 2)):
            gpu_ctx.set_playouts_per_lane(ppl)
            gpu_ctx.set_regroup(rounds, below, shrink)
            for prep in (False, True):
                q = gpu_ctx.rollout(b, d, r, p, max_steps=1000, prep=prep, return_state=True)
                ref = plain
                if prep:
                    gpu_ctx.set_playouts_per_lane(1)
                    ref = gpu_ctx.rollout(b, d, r, p, max_steps=1000, prep=True, return_state=True)
                    gpu_ctx.set_playouts_per_lane(ppl)
                for key in ("results", "steps", "values", "battles", "durations", "prng"):
                    assert (q[key] == ref[key]).all(), (ppl, rounds, below, shrink, prep, key)
    finally:
        gpu_ctx.set_playouts_per_lane(2)
        gpu_ctx.set_regroup()


def test_long_playout_migration_does_not_change_results(gpu_ctx):
    """k_rollout_queue's in-launch migration (bulk waves hand playouts that run beyond `long_steps` to adopter waves through
    the parked state image + an agent-scope release / acquire): forced on for small launches with thresholds that make
    MANY playouts migrate; every output byte equals the plain launch's and the oracle's, every donation is adopted, no
    bounded wait ran out."""
    n = 6000
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x7E57AB1E)
    ob, od, op = b.copy(), d.copy(), p.copy()
    oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, threads=8)
    gpu_ctx.set_playouts_per_lane(1)
    gpu_ctx.set_migration(0)
    plain = gpu_ctx.rollout(b, d, r, p, max_steps=1000, return_state=True)
    assert (plain["steps"] == osteps).all() and (plain["battles"] == ob).all()
    try:
        # (window: the standstill counter of round 4 -- a playout whose actives kept their slots and hp for that many turn-steps is
        # donated too; 0 = off.  Windows of 1-3 turn-steps donate nearly every playout several times over.)
        for ppl, long_steps, adopters, window in ((2, 20, 2, 0), (2, 60, 1, 0), (3, 5, 4, 0), (2, 150, 8, 0), (4, 1, 3, 0),
                                                  (2, 999, 4, 2), (2, 300, 8, 8), (3, 999, 2, 1), (2, 999, 16, 40)):
            gpu_ctx.set_playouts_per_lane(ppl)
            gpu_ctx.set_migration(2, long_steps, adopters)
            gpu_ctx.set_migration_window(window)
            for prep in (False, True):
                q = gpu_ctx.rollout(b, d, r, p, max_steps=1000, prep=prep, return_state=True)
                c = gpu_ctx.queue_counters()
                assert c[63] == 0, ("bounded wait ran out", ppl, long_steps, adopters, int(c[63]))
                assert c[40] == c[41] and c[40] > 0, ("donations / adoptions", int(c[40]), int(c[41]))
                ref = plain
                if prep:
                    gpu_ctx.set_playouts_per_lane(1)
                    gpu_ctx.set_migration(0)
                    ref = gpu_ctx.rollout(b, d, r, p, max_steps=1000, prep=True, return_state=True)
                    gpu_ctx.set_playouts_per_lane(ppl)
                    gpu_ctx.set_migration(2, long_steps, adopters)
                for key in ("results", "steps", "values", "battles", "durations", "prng"):
                    assert (q[key] == ref[key]).all(), (ppl, long_steps, adopters, prep, key)
    finally:
        gpu_ctx.set_playouts_per_lane(2)
        gpu_ctx.set_migration(1, 300, 0)
        gpu_ctx.set_migration_window(48)


def test_two_contexts_migrate_concurrently_on_one_device(gpu_ctx):
    """Co-residency is not a correctness requirement of k_rollout_queue's migration: a bulk wave never waits for anyone,
    so it always finishes and counts itself out; adopters only ever wait for bulk waves (bounded) and hold nothing a bulk wave needs.
    Two contexts launch saturating group launches AT THE SAME TIME on their own streams -- each grid alone would fill the device's wave
    slots, so neither is fully resident while the other runs -- with migration forced on and thresholds that donate thousands of playouts:
    every output byte of both launches equals the oracle's, donations == adoptions, and both sticky error words are 0."""
    from hipmem import Dev
    from oak_amd import _lib
    from oak_amd.engine import Context
    nb, n = 5, 65536                      # 5 batches x 65,536 playouts per context: 327,680 > the 262,144 resident lanes
    other = Context(0)
    ctxs = [gpu_ctx, other]
    runs = []
    try:
        for k, ctx in enumerate(ctxs):
            ctx.set_playouts_per_lane(2)
            ctx.set_migration(2, 120, 0)          # forced on; donate everything still running after 120 turn-steps
            ctx.set_migration_window(24)
            b, d, p, r = O.make_random_ou_batch(nb * n, seed0=0xC0DE0000 + k * nb * n)
            dev = {"b": Dev(b), "d": Dev(d), "p": Dev(p), "r": Dev(r), "ro": Dev(r, fill=0), "st": Dev(np.zeros(nb * n, dtype=np.uint32)),
                   "va": Dev(np.zeros(nb * n, dtype=np.float32)), "bo": Dev(b, fill=0), "do": Dev(d, fill=0)}
            batches = (_lib.RolloutBatch * nb)()
            for j in range(nb):
                o = j * n
                at = lambda key, stride: C.c_void_p(dev[key].p.value + o * stride)
                batches[j] = _lib.RolloutBatch(at("b", 384), at("d", 8), at("r", 1), at("p", 8), n, at("ro", 1), at("st", 4), at("va", 4), at("bo", 384), at("do", 8))
            runs.append((ctx, dev, batches, (b, d, p, r)))
        for rep in range(2):                       # both launches in flight together, twice
            for ctx, dev, batches, _ in runs:
                _lib.check(ctx.lib.oakgpu_rollout_group_dev(ctx.handle, batches, nb, 1000, 0))
        for ctx, dev, batches, (b, d, p, r) in runs:
            c = ctx.queue_counters()              # synchronises the context's stream
            assert c[63] == 0, ("bounded wait ran out", int(c[63]))
            assert c[40] == c[41] and c[40] > 1000, ("donations / adoptions", int(c[40]), int(c[41]))
            # the second launch started from the first one's choice streams (prng advances in place): replay both on the oracle
            ob, od, op = b.copy(), d.copy(), p.copy()
            O.rollout_batch(ob, od, r, op, max_steps=1000, threads=16)
            ob, od = b.copy(), d.copy()
            oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, threads=16)
            assert (dev["ro"].host() == oout).all() and (dev["st"].host() == osteps).all()
            assert (dev["bo"].host() == ob).all() and (dev["do"].host() == od).all() and (dev["p"].host() == op).all()
    finally:
        for ctx, dev, _, _ in runs:
            try:
                ctx.synchronize()
            except Exception:
                pass
            for x in dev.values():
                x.free()
        gpu_ctx.set_migration(1, 300, 0)
        gpu_ctx.set_migration_window(48)
        other.close()


def test_rollout_in_place_on_device_buffers(gpu_ctx):
    """oakgpu_rollout_dev with battles_out == battles, durations_out == durations, results_out == results_in
    (how bench.py --workload config3 steps a resident batch one turn at a time): same bytes as out of place.
    Device buffers come straight from the HIP runtime the library already loaded (no torch in this process)."""
    import ctypes as C
    from oak_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    H2D, D2H = 1, 2

    class Dev:
        def __init__(self, arr):
            self.shape, self.dtype, self.nbytes = arr.shape, arr.dtype, arr.nbytes
            self.p = C.c_void_p()
            assert hip.hipMalloc(C.byref(self.p), self.nbytes) == 0
            a = np.ascontiguousarray(arr)
            assert hip.hipMemcpy(self.p, a.ctypes.data_as(C.c_void_p), self.nbytes, H2D) == 0

        def host(self):
            out = np.empty(self.shape, dtype=self.dtype)
            assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.p, self.nbytes, D2H) == 0
            return out

        def free(self):
            hip.hipFree(self.p)
    n = 5000
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x1A2B3C)
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    for max_steps, turns in ((1, 6), (1000, 1)):
        gb, gd, gp, gr = Dev(b), Dev(d), Dev(p), Dev(r)
        steps, vals = Dev(np.zeros(n, dtype=np.uint32)), Dev(np.zeros(n, dtype=np.float32))
        hb, hd, hp, hr = b.copy(), d.copy(), p.copy(), r.copy()
        for _ in range(turns):
            _lib.check(lib.oakgpu_rollout_dev(h, gb.p, gd.p, gr.p, gp.p, n, max_steps, 0, gr.p, steps.p, vals.p, gb.p, gd.p))
            gpu_ctx.synchronize()
            ref = gpu_ctx.rollout(hb, hd, hr, hp, max_steps=max_steps, return_state=True)
            hb, hd, hp, hr = ref["battles"], ref["durations"], ref["prng"], ref["results"]
            assert (gb.host() == hb).all() and (gd.host() == hd).all()
            assert (gr.host() == hr).all() and (gp.host() == hp).all()
            assert (steps.host() == ref["steps"]).all() and (vals.host() == ref["values"]).all()
        for x in (gb, gd, gp, gr, steps, vals):
            x.free()


def test_libpkmn_named_single_battle_abi(gpu_ctx):
    """include/pkmn.h: pkmn_gen1_battle_update / _choices / _options_* as batches of one on the GPU,
    driven exactly like the reference's rollout loop (mcts.h:448-496) and compared with the oracle."""
    import ctypes as C
    from oak_amd import _lib
    lib = _lib.load()

    class Options(C.Structure):
        _fields_ = [("actions", C.c_uint8 * 16), ("durations", C.c_uint8 * 8), ("overrides", C.c_uint8 * 16), ("has", C.c_uint8)]
    lib.pkmn_gen1_battle_update.restype = C.c_uint8
    lib.pkmn_gen1_battle_update.argtypes = [C.c_void_p, C.c_uint8, C.c_uint8, C.c_void_p]
    lib.pkmn_gen1_battle_choices.restype = C.c_uint8
    lib.pkmn_gen1_battle_choices.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
    lib.pkmn_gen1_battle_options_set.argtypes = [C.c_void_p] * 4
    lib.pkmn_result_type.restype = C.c_int
    lib.pkmn_result_type.argtypes = [C.c_uint8]
    b, d, p, r = O.make_random_ou_batch(6, seed0=0xBEEF)

    def play(i, seed):
        rng = np.random.default_rng(seed)
        gb = b[i].copy()
        opt, oopt = Options(), O.Options()
        res = ores = int(r[i])
        out = (C.c_uint8 * 9)()
        for _ in range(40):
            if lib.pkmn_result_type(res) != 0:
                break
            picks = []
            for pl in (0, 1):
                req = (res >> (4 + 2 * pl)) & 3
                n = lib.pkmn_gen1_battle_choices(O.ptr(gb), pl, req, out, 9)
                oc = O.choices(b[i], pl, req)
                assert n == len(oc) and list(out[:n]) == list(oc)
                picks.append(int(oc[rng.integers(n)]))
            lib.pkmn_gen1_battle_options_set(C.byref(opt), None, None, None)
            res = lib.pkmn_gen1_battle_update(O.ptr(gb), picks[0], picks[1], C.byref(opt))
            oopt.set()
            ores = O.update(b[i], picks[0], picks[1], oopt)
            assert res == ores and (gb == b[i]).all()
            assert bytes(opt.durations) == oopt.durations.tobytes() and bytes(opt.actions) == oopt.actions.tobytes()
        return True

    for i in range(3):
        play(i, 1)
    # libpkmn is thread-safe per battle and the reference runs one game per thread (generate.cc:527-536): three threads drive
    # three battles through the libpkmn-named ABI at once -- each on its own per-thread context -- and every step still
    # agrees with the oracle
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=3) as ex:
        assert all(f.result() for f in [ex.submit(play, 3 + t, 10 + t) for t in range(3)])


def test_full_size_config2_bit_exact(gpu_ctx):
    """BASELINE configs[1] at its full size: all 65,536 random OU playouts, every output byte vs the oracle."""
    n = 65536
    b, d, p, r = O.make_random_ou_batch(n)
    got = gpu_ctx.rollout(b, d, r, p, max_steps=1000, return_state=True)   # default schedule = the one bench.py uses
    ob, od, op = b.copy(), d.copy(), p.copy()
    oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, threads=16)
    assert (got["steps"] == osteps).all() and (got["results"] == oout).all()
    assert (got["battles"] == ob).all() and (got["durations"] == od).all() and (got["prng"] == op).all()
    assert int(osteps.sum()) > 6_000_000


def test_full_size_config4_properties_and_sampled_parity(gpu_ctx):
    """BASELINE config 4 at full size (256 roots x 4096 replicas = 1,048,576 playouts, device-side prep):
    size-independent properties on everything, bit-exact parity on a random 8,192-lane sample (lanes are
    independent, so a lane's outputs depend only on its own inputs), determinism across launches."""
    roots, reps = 256, 4096
    b, d, p, r = O.make_random_ou_batch(roots, seed0=0x0A4B00000000)     # first 256 lanes of config 2
    n = roots * reps
    B, D, R = np.repeat(b, reps, axis=0), np.repeat(d, reps, axis=0), np.repeat(r, reps)
    rng = np.random.default_rng(4)
    prng = rng.integers(0, 256, (n, 8), dtype=np.uint8)
    prng[:, 0] |= 1                                                      # never the all-zero state
    got = gpu_ctx.rollout(B, D, R, prng, max_steps=1000, prep=True, return_state=True)
    again = gpu_ctx.rollout(B, D, R, prng, max_steps=1000, prep=True)
    assert (again["results"] == got["results"]).all() and (again["steps"] == got["steps"]).all()
    t = got["results"] & 15
    assert ((t <= 3)).all() and (got["steps"] <= 1000).all() and (got["steps"][t == 0] == 1000).all()
    assert set(np.unique(got["values"])) <= {0.0, 0.5, 1.0}
    per_root = got["values"].reshape(roots, reps).mean(axis=1)
    assert ((per_root >= 0.0) & (per_root <= 1.0)).all() and 0.4 < float(per_root.mean()) < 0.6   # random teams: no side bias
    pick = rng.choice(n, 8192, replace=False)
    ob, od, op = B[pick].copy(), D[pick].copy(), prng[pick].copy()
    oout, osteps = O.rollout_batch(ob, od, R[pick], op, max_steps=1000, prep=True, threads=16)
    assert (got["results"][pick] == oout).all() and (got["steps"][pick] == osteps).all()
    assert (got["battles"][pick] == ob).all() and (got["durations"][pick] == od).all() and (got["prng"][pick] == op).all()


def test_group_launch_matches_separate_launches(gpu_ctx):
    """oakgpu_rollout_group: several batches (ragged sizes, an empty one, a one-playout one) drained through ONE playout
    queue give the byte-identical outputs of separate launches and of the oracle, with and without root prep, and
    under regrouping rounds (suspended playouts of different batches share the scratch lists)."""
    sizes = (1500, 0, 1, 777, 64, 2049)
    batches = [O.make_random_ou_batch(n, seed0=0x6A0F0000 + 7919 * k) if n else
               (np.zeros((0, 384), np.uint8), np.zeros((0, 8), np.uint8), np.zeros((0, 8), np.uint8), np.zeros(0, np.uint8))
               for k, n in enumerate(sizes)]
    try:
        for ppl, rounds, below, shrink, max_steps in ((2, 1, 0, 1, 1000), (3, 4, 32, 3, 1000), (2, 3, 64, 2, 40)):
            gpu_ctx.set_playouts_per_lane(ppl)
            gpu_ctx.set_regroup(rounds, below, shrink)
            for prep in (False, True):
                got = gpu_ctx.rollout_group([(b, d, r, p) for b, d, p, r in batches], max_steps=max_steps, prep=prep, return_state=True)
                for (b, d, p, r), g in zip(batches, got):
                    if b.shape[0] == 0:
                        assert g["steps"].size == 0
                        continue
                    ob, od, op = b.copy(), d.copy(), p.copy()
                    oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=max_steps, prep=prep, threads=4)
                    assert (g["steps"] == osteps).all() and (g["results"] == oout).all()
                    assert (g["battles"] == ob).all() and (g["durations"] == od).all() and (g["prng"] == op).all()
                    t = oout & 15
                    assert (g["values"] == np.where(t == 1, 1.0, np.where(t == 2, 0.0, 0.5)).astype(np.float32)).all()
    finally:
        gpu_ctx.set_playouts_per_lane(2)
        gpu_ctx.set_regroup()


def test_config1_benchmark_playouts_on_the_gpu(gpu_ctx):
    """BASELINE configs[0]: the two benchmark teams, 1,000 playouts driven by ONE shared std::mt19937{1111111}
    (benchmark.cc:23-31 + mcts.h:250-263,448-496), through the HIP path: the generator's output is produced on the host
    (oakgpu_mt19937_fill), start offsets are resolved on the device, and the (turn-steps, P1 wins, final-state digest)
    triple must equal tests/golden/config1_digest.json byte for byte -- the same file the CPU oracle is held to."""
    import json
    import os
    from oak_amd.engine import mt19937_uniform_64
    from test_oracle_goldens import benchmark_teams, ROOT
    teams = np.array(benchmark_teams(), dtype=np.uint8).reshape(1, 2, 6, 5)
    b0, d0, r0 = gpu_ctx.battle(teams, np.array([1111111], dtype=np.uint64), first_update=True)   # PKMN::battle + update(0, 0)
    ob = O.init_battle(benchmark_teams(), 1111111)
    assert O.update(ob, 0, 0, O.Options()) == r0[0] and (ob == b0[0]).all()
    draws = mt19937_uniform_64(1111111, 200000)
    got = gpu_ctx.rollout_shared_device(b0[0], d0[0], int(r0[0]), draws, 1000, max_steps=100000, prep=True, return_state=True)
    digest = 0xcbf29ce484222325
    for i in range(1000):
        digest = (digest ^ O.LIB.oracle_hash64(O.ptr(got["battles"][i]), 384)) * 0x100000001b3 % 2**64
    triple = {"turn_steps": int(got["steps"].sum()), "p1_wins": int(((got["results"] & 15) == 1).sum()), "digest": "%016x" % digest}
    assert triple == json.load(open(os.path.join(ROOT, "tests", "golden", "config1_digest.json")))
    assert got["consumed"] == 1000 + triple["turn_steps"]          # one draw for battle.rng + one per turn-step
    assert got["offsets"][0] == 0 and (np.diff(got["offsets"].astype(np.int64)) == got["steps"][:-1].astype(np.int64) + 1).all()
    # a stream that is too short is refused, not truncated
    with pytest.raises(RuntimeError):
        gpu_ctx.rollout_shared_device(b0[0], d0[0], int(r0[0]), draws[:5000], 1000, max_steps=100000, prep=True)


def test_rollout_draws_per_lane_roots_match_oracle(gpu_ctx):
    """oakgpu_rollout_draws_dev with a different root per lane and explicit offsets: every lane equals the oracle's
    mt19937-driven rollout started at that offset of the same stream (no prep, capped)."""
    from oak_amd import _lib
    from oak_amd.engine import mt19937_uniform_64
    n = 300
    b, d, p, r = O.make_random_ou_batch(n, seed0=0xD2A35000)
    draws = mt19937_uniform_64(77, 4096)
    offsets = (np.arange(n, dtype=np.uint32) * 7) % 3000
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]

    def dev(a):
        ptr = C.c_void_p()
        assert hip.hipMalloc(C.byref(ptr), max(a.nbytes, 4)) == 0
        assert hip.hipMemcpy(ptr, np.ascontiguousarray(a).ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
        return ptr

    def host(ptr, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), ptr, out.nbytes, 2) == 0
        return out
    outs = dict(res=np.zeros(n, np.uint8), steps=np.zeros(n, np.uint32), vals=np.zeros(n, np.float32), bo=np.zeros((n, 384), np.uint8),
                do=np.zeros((n, 8), np.uint8), used=np.zeros(n, np.uint32))
    ptrs = {k: dev(v) for k, v in outs.items()}
    gb, gd, gr, gdr, goff = dev(b), dev(d), dev(r), dev(draws), dev(offsets)
    cap = 150
    _lib.check(gpu_ctx.lib.oakgpu_rollout_draws_dev(gpu_ctx.handle, gb, 384, gd, 8, gr, 1, gdr, len(draws), goff, n, cap, 0, ptrs["res"],
                                                     ptrs["steps"], ptrs["vals"], ptrs["bo"], ptrs["do"], ptrs["used"]))
    gpu_ctx.synchronize()
    got = {k: host(ptrs[k], v.shape, v.dtype) for k, v in outs.items()}
    for x in list(ptrs.values()) + [gb, gd, gr, gdr, goff]:
        hip.hipFree(x)
    for i in range(n):
        ob, od = b[i].copy(), d[i].copy()
        opt = O.Options(od)
        res, k = int(r[i]), 0
        while (res & 15) == 0 and k < cap:
            seed = int(draws[offsets[i] + k])
            c1s, c2s = O.choices(ob, 0, (res >> 4) & 3), O.choices(ob, 1, (res >> 6) & 3)
            opt.set()
            res = O.update(ob, int(c1s[seed % len(c1s)]), int(c2s[(seed >> 32) % len(c2s)]), opt)
            k += 1
        assert got["steps"][i] == k and got["used"][i] == k and got["res"][i] == res, i
        assert (got["bo"][i] == ob).all() and (got["do"][i] == opt.durations).all(), i


def test_set_ou_pools_rejects_inputs_that_would_hang_the_generator():
    """k_random_ou picks 6 distinct species per side and distinct moves per species by rejection: fewer than 6 legal
    species, duplicates, or a pool with repeated moves would spin forever on the GPU -- the ABI refuses them."""
    from oak_amd import _lib, gamedata
    from oak_amd.engine import Context
    ctx = Context(0)
    legal, pools, sizes = gamedata.ou_pools()
    pools = np.ascontiguousarray(pools)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    call = lambda l, n, p, s: ctx.lib.oakgpu_set_ou_pools(ctx.handle, P(l), n, P(p), P(s))
    assert call(legal, len(legal), pools, sizes) == 0
    assert call(legal, 5, pools, sizes) != 0                       # fewer than 6 species
    dup = legal.copy(); dup[3] = dup[2]
    assert call(dup, len(dup), pools, sizes) != 0                  # duplicate species
    bad_pool = pools.copy(); bad_pool[int(legal[0]), 1] = bad_pool[int(legal[0]), 0]
    if sizes[int(legal[0])] >= 2:
        assert call(legal, len(legal), bad_pool, sizes) != 0       # repeated move in a pool
    zero = sizes.copy(); zero[int(legal[1])] = 0
    assert call(legal, len(legal), pools, zero) != 0               # empty pool
    ctx.close()


def test_segment_mean_and_direct_rccl_all_gather(gpu_ctx):
    """The C++ host layer's exchange step: per-root means on the device (oakgpu_segment_mean_dev) and the library's own
    ncclAllGather call site (oakgpu_comm_* / oakgpu_all_gather_dev) -- here with a one-rank communicator on the one GPU."""
    from oak_amd import _lib
    hip = C.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    rng = np.random.default_rng(3)
    for segs, per in ((256, 4096), (7, 100), (1, 1), (32, 333)):
        vals = rng.random((segs, per)).astype(np.float32)
        dv, do = C.c_void_p(), C.c_void_p()
        assert hip.hipMalloc(C.byref(dv), vals.nbytes) == 0 and hip.hipMalloc(C.byref(do), segs * 4) == 0
        assert hip.hipMemcpy(dv, vals.ctypes.data_as(C.c_void_p), vals.nbytes, 1) == 0
        _lib.check(lib.oakgpu_segment_mean_dev(h, dv, segs, per, do))
        gpu_ctx.synchronize()
        out = np.zeros(segs, dtype=np.float32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), do, segs * 4, 2) == 0
        assert np.abs(out - vals.astype(np.float64).mean(axis=1)).max() <= 2e-6
        hip.hipFree(dv)
        hip.hipFree(do)
    ident = (C.c_uint8 * 128)()
    _lib.check(lib.oakgpu_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(lib.oakgpu_comm_create(h, ident, 0, 1, C.byref(comm)))
    send = rng.random(256).astype(np.float32)
    ds, dr = C.c_void_p(), C.c_void_p()
    assert hip.hipMalloc(C.byref(ds), 1024) == 0 and hip.hipMalloc(C.byref(dr), 1024) == 0
    assert hip.hipMemcpy(ds, send.ctypes.data_as(C.c_void_p), 1024, 1) == 0
    _lib.check(lib.oakgpu_all_gather_dev(h, comm, ds, dr, 256))
    gpu_ctx.synchronize()
    recv = np.zeros(256, dtype=np.float32)
    assert hip.hipMemcpy(recv.ctypes.data_as(C.c_void_p), dr, 1024, 2) == 0
    assert (recv == send).all()
    lib.oakgpu_comm_destroy(comm)
    hip.hipFree(ds)
    hip.hipFree(dr)


def test_lds_resident_engine_rollouts_are_bit_identical(gpu_ctx):
    """Engine 1 (k_rollout<64>: gen1_device.hpp's LDS-resident engine, written first and kept as a second implementation
    of the same turn resolution) against the oracle: every output byte, with and without root prep, capped, ragged, and as
    a group launch.  (The register-resident engine is what every other rollout test runs.)"""
    try:
        gpu_ctx.set_rollout_engine(1)
        for n, max_steps, seed0 in ((3000, 1000, 0xB1A50000), (700, 23, 0xB1A51000), (9000, 1000, 0xB1A52000), (1, 1000, 0xB1A53000)):
            b, d, p, r = O.make_random_ou_batch(n, seed0=seed0)
            for prep in (False, True):
                got = gpu_ctx.rollout(b, d, r, p, max_steps=max_steps, prep=prep, return_state=True)
                ob, od, op = b.copy(), d.copy(), p.copy()
                oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=max_steps, prep=prep, threads=8)
                assert (got["steps"] == osteps).all() and (got["results"] == oout).all(), (n, prep)
                bad = np.nonzero((got["battles"] != ob).any(axis=1))[0]
                assert bad.size == 0, (n, prep, int(bad[0]))
                assert (got["durations"] == od).all() and (got["prng"] == op).all()
        batches = [O.make_random_ou_batch(n, seed0=0xB1A60000 + 1000 * k) for k, n in enumerate((1200, 77, 2500))]
        got = gpu_ctx.rollout_group([(b, d, r, p) for b, d, p, r in batches], return_state=True)
        for (b, d, p, r), g in zip(batches, got):
            ob, od, op = b.copy(), d.copy(), p.copy()
            oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, threads=8)
            assert (g["steps"] == osteps).all() and (g["results"] == oout).all() and (g["battles"] == ob).all() and (g["prng"] == op).all()
        with pytest.raises(RuntimeError, match="engine must be 1"):
            gpu_ctx.set_rollout_engine(3)
    finally:
        gpu_ctx.set_rollout_engine(2)


def test_group_launch_limits(gpu_ctx):
    """64 batches per group is the documented maximum (the batch table lives in LDS): 64 tiny batches work and agree with
    the oracle, 65 are refused with a message, and a null required pointer is refused before anything is launched."""
    from oak_amd import _lib
    batches = [O.make_random_ou_batch(3 + (k % 5), seed0=0x64000000 + 97 * k) for k in range(64)]
    got = gpu_ctx.rollout_group([(b, d, r, p) for b, d, p, r in batches], max_steps=300, return_state=True)
    for (b, d, p, r), g in zip(batches, got):
        ob, od, op = b.copy(), d.copy(), p.copy()
        oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=300)
        assert (g["steps"] == osteps).all() and (g["results"] == oout).all() and (g["battles"] == ob).all()
    with pytest.raises(RuntimeError, match="more than 64 batches"):
        gpu_ctx.rollout_group([(b, d, r, p) for b, d, p, r in batches] + [(batches[0][0], batches[0][1], batches[0][3], batches[0][2])])
    descs = (_lib.RolloutBatch * 1)()
    descs[0] = _lib.RolloutBatch(None, None, None, None, 5, None, None, None, None, None)
    assert gpu_ctx.lib.oakgpu_rollout_group_dev(gpu_ctx.handle, descs, 1, 10, 0) != 0
    assert b"null required pointer" in gpu_ctx.lib.oakgpu_last_error()


def test_root_groups_pipeline_gives_the_unpipelined_per_root_results():
    """configs[3] as independent root groups (oak_amd.dist.RootGroups): 12 roots x 256 playouts with root prep,
    two search steps, cut into 1 / 3 / 5 (ragged) groups on their own contexts, launched in fixed order and work-conserving:
    every playout's result, step count and choice-RNG state after both steps, and every root's mean of either step, equal the
    oracle's (= the unpipelined step's) -- scheduling never shows in the results.  Runs tests/root_groups_check.py in a child
    process: RootGroups keeps its buffers in torch tensors, and torch must initialise the GPU before the library does."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "root_groups_check.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "root groups ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def _frozen_standstills(n, seed0, clear_volatiles_every=2, turn=None):
    """Mid-game random OU battles patched into last-Pokemon-against-last-Pokemon positions with BOTH actives frozen -- the stalemate
    k_rollout_queue takes in one go (EngineR::frozen_standstill).  Every `clear_volatiles_every`-th lane has its actives' volatiles
    cleared (the proven case); the others keep whatever the game left there (Leech Seed, confusion, substitutes ...: mostly NOT
    the proven case, same answer required)."""
    b, d, p, r = O.make_random_ou_batch(n, seed0=seed0)
    res, _ = O.rollout_batch(b, d, r, p, max_steps=24, threads=8)
    keep = np.where(res == 0x50)[0]          # still running, both sides asked for a move
    b, d, p, res = b[keep].copy(), d[keep].copy(), p[keep].copy(), res[keep].copy()
    for k in range(len(keep)):
        for s in range(2):
            so = s * 184
            act = int(b[k, so + 176]) - 1      # order[0]: the active's party slot
            locked = k % 8 in (1, 5) and s == (k % 8 == 5)   # a side that keeps its bench but is LOCKED into a move (Rage / recharging)
            for q in range(6):
                o = so + q * 24
                if q != act:
                    if not locked:
                        b[k, o + 18] = b[k, o + 19] = 0     # hp 0: fainted
                        b[k, o + 20] = 0
                else:
                    b[k, o + 20] = 0x20                 # FRZ
                    if b[k, o + 18] == 0 and b[k, o + 19] == 0:
                        b[k, o + 18] = 1
            if k % clear_volatiles_every == 0 or locked:
                b[k, so + 144 + 16:so + 144 + 24] = 0   # the active's volatiles
            if locked:
                b[k, so + 144 + 17] = 0x10 if k % 8 == 1 else 0x08   # V_RAGE (bit 12) / V_RECHARGING (bit 11)
        if turn is not None:
            b[k, 368], b[k, 369] = turn & 0xFF, turn >> 8
    d[:] = 0
    return b, d, p, res


@pytest.mark.gpu
def test_frozen_standstill_skip_is_exact(gpu_ctx):
    """k_rollout_queue takes a PROVEN standstill (both sides' last Pokemon frozen, nothing acting on them, different speeds) to its last
    turn-step in one go; the oracle plays every turn.  Same bytes: by step cap, by the 1,000-turn tie, from late turns, with volatiles
    that do and do not satisfy the proof, and with equal speeds (no skip: a tie is drawn every turn)."""
    gpu_ctx.set_playouts_per_lane(2)
    for seed0, cap, turn in ((0xF0F0_0001, 1000, None), (0xF0F0_0002, 137, None), (0xF0F0_0003, 1000, 960), (0xF0F0_0004, 50, 998), (0xF0F0_0005, 1000, None)):
        b, d, p, r = _frozen_standstills(3000, seed0, turn=turn)
        if seed0 == 0xF0F0_0001:                      # a tenth of the lanes: equal speeds (bytes 6-7 of the active's stats)
            for k in range(0, len(b), 10):
                b[k, 184 + 144 + 6:184 + 144 + 8] = b[k, 144 + 6:144 + 8]
        if seed0 == 0xF0F0_0005:                      # every combination of the volatile FLAGS (bits 0-16; counters stay 0) on both actives:
            rs = np.random.RandomState(5)             # all branches of the proof's condition -- Bide / Rage / binding / Leech Seed / ...
            for k in range(len(b)):
                for so in (0, 184):
                    m = int(rs.randint(0, 1 << 17)) if k % 3 else int(1 << rs.randint(0, 17))
                    b[k, so + 144 + 16:so + 144 + 24] = 0
                    b[k, so + 144 + 16], b[k, so + 144 + 17], b[k, so + 144 + 18] = m & 0xFF, (m >> 8) & 0xFF, (m >> 16) & 1
                    # a lock goes with the move that causes it: a side "thrashing" in Toxic is a state no game reaches, and once a Haze
                    # thaws it the engines are free to differ there (the checker resolves status moves before the lock check, the
                    # register engine behind it -- seen in a variant build, round 5)
                    for bit, move in ((1, 37), (4, 76), (0, 117), (12, 99)):      # Thrash, SolarBeam, Bide, Rage
                        if m >> bit & 1:
                            b[k, so + 182] = move
                            break
        assert len(b) > 1000
        ob, od, op = b.copy(), d.copy(), p.copy()
        oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=cap, threads=8)
        got = gpu_ctx.rollout(b, d, r, p, max_steps=cap, return_state=True)
        assert (((oout & 15) == 3) | (osteps >= cap)).mean() > (0.3 if seed0 == 0xF0F0_0005 else 0.9), "the fixture no longer produces standstills"
        for key, exp in (("results", oout), ("steps", osteps), ("battles", ob), ("durations", od), ("prng", op)):
            assert (got[key] == exp).all(), (hex(seed0), cap, turn, key, int((got[key] != exp).sum()))
        if seed0 == 0xF0F0_0002:                      # the runtime switch: every turn-step played, same bytes (round-4 advice: an A/B handle)
            gpu_ctx.set_standstill_skip(False)
            try:
                slow = gpu_ctx.rollout(b, d, r, p, max_steps=cap, return_state=True)
            finally:
                gpu_ctx.set_standstill_skip(True)
            for key in ("results", "steps", "battles", "durations", "prng"):
                assert (slow[key] == got[key]).all(), key


# ---- BASELINE configs[3] in slices: oakgpu_root_steps (k_root_step) against the oracle that applies the same crediting rule ----------
def _root_step_inputs(n_roots, seed0):
    """Roots at different depths of their game: turn-1 positions, mid-game positions with live durations (so that root prep -- the
    hidden-variable resampling of durations.h:25-97 -- matters), and one finished game (a playout of length 0)."""
    b, d, p, r = O.make_random_ou_batch(n_roots, seed0=seed0)
    for i in range(n_roots):
        k = (0, 0, 7, 19, 33, 60)[i % 6]
        if k:
            out, _ = O.rollout_batch(b[i:i + 1], d[i:i + 1], r[i:i + 1], p[i:i + 1], max_steps=k)
            r[i] = out[0]
    out, _ = O.rollout_batch(b[-1:], d[-1:], r[-1:], p[-1:], max_steps=1000)     # the last root: a terminal position
    r[-1] = out[0]
    assert (r[-1] & 15) != 0
    return b, d, r


class _RootSteps:
    """The C ABI of the sliced search steps driven with raw device buffers (no torch in this process: tests/hipmem.py)."""

    def __init__(self, ctx, b, d, r, lane, reps, slice_, max_steps=1000):
        from hipmem import Dev
        from oak_amd import _lib
        self.ctx, self.lib, self.n = ctx, ctx.lib, b.shape[0]
        self.bufs = [Dev(np.ascontiguousarray(x)) for x in (b, d, r, lane)]
        self.report = Dev(np.zeros(self.n + 2, dtype=np.uint64))
        self.h = C.c_void_p()
        _lib.check(self.lib.oakgpu_root_steps_create(ctx.handle, self.n, reps, slice_, max_steps, C.byref(self.h)))

    def step(self, fresh=True):
        from oak_amd import _lib
        tb, td, tr, tp = self.bufs
        _lib.check(self.lib.oakgpu_root_steps_launch_dev(self.h, tb.p, td.p, tr.p, tp.p, 1 if fresh else 0, self.report.p))
        self.ctx.synchronize()
        rep = self.report.host()
        acc = rep[:self.n]
        return {"count": (acc & np.uint64(0xFFFFFFFF)).astype(np.int64), "sum2": (acc >> np.uint64(32)).astype(np.int64),
                "turn_steps": int(rep[self.n]), "carried": int(rep[self.n + 1] & np.uint64(0xFFFFFFFF)), "err": int(rep[self.n + 1] >> np.uint64(32))}

    def lane_streams(self):
        return self.bufs[3].host()

    def close(self):
        self.lib.oakgpu_root_steps_destroy(self.h)
        for x in self.bufs + [self.report]:
            x.free()


@pytest.mark.parametrize("slice_,reps,steps", [(64, 96, 5), (16, 64, 4), (256, 64, 3), (0, 64, 2)])
def test_root_steps_credit_each_playout_by_its_own_length(gpu_ctx, slice_, reps, steps):
    """Every step's per-root aggregate (count, 2 x value sum) of the sliced search steps equals the oracle's, which plays every playout to
    terminal at once and applies the documented rule (step k + (len - 1) // slice); so do the turn-steps each launch executes, the
    drain steps' aggregates, the lane streams afterwards, and -- after the drain -- nothing is left in flight."""
    n_roots = 13
    b, d, r = _root_step_inputs(n_roots, 0xC0FFEE00 + slice_)
    lane = _seed_prng(n_roots * reps, 0xC40000000000)
    ref_lane = lane.copy()
    cnt, s2, ex = O.root_steps_reference(b, d, r, ref_lane, reps, steps, slice_, threads=8)
    rs = _RootSteps(gpu_ctx, b, d, r, lane, reps, slice_)
    try:
        k = 0
        while True:
            rec = rs.step(fresh=k < steps)
            assert rec["err"] == 0
            assert (rec["count"] == cnt[k]).all(), (k, rec["count"], cnt[k])
            assert (rec["sum2"] == s2[k]).all(), k
            assert rec["turn_steps"] == ex[k], (k, rec["turn_steps"], ex[k])
            k += 1
            if k >= steps and rec["carried"] == 0:
                break
            assert k < cnt.shape[0], "playouts still in flight after the longest possible life"
        assert cnt[k:].sum() == 0 and cnt[:k].sum() == n_roots * reps * steps      # every playout credited exactly once
        assert (rs.lane_streams() == ref_lane).all()                                 # one uniform_64 per lane and step
        if slice_ == 0:
            assert k == steps
    finally:
        rs.close()


def test_root_steps_do_not_depend_on_the_partition_of_the_roots(gpu_ctx):
    """Two objects over disjoint parts of the roots (what two ranks hold: lane streams seeded by GLOBAL lane index) credit exactly what
    one object over all roots credits -- nothing depends on which launch, wave or rank runs a playout."""
    from oak_amd.engine import Context
    n_roots, reps, steps, slice_ = 12, 64, 4, 32
    b, d, r = _root_step_inputs(n_roots, 0xD15C0)
    lane = _seed_prng(n_roots * reps, 0xC40000000000)

    def run(ctx, lo, hi):
        rs = _RootSteps(ctx, b[lo:hi], d[lo:hi], r[lo:hi], lane[lo * reps:hi * reps], reps, slice_)
        out = []
        for k in range(steps + 1000 // slice_ + 1):
            rec = rs.step(fresh=k < steps)
            out.append(np.stack([rec["count"], rec["sum2"]]))
        assert rec["carried"] == 0 and rec["err"] == 0
        rs.close()
        return np.stack(out)            # [step, 2, roots]
    whole = run(gpu_ctx, 0, n_roots)
    other = Context(0)
    try:
        parts = np.concatenate([run(gpu_ctx, 0, 5), run(other, 5, n_roots)], axis=2)
    finally:
        other.close()
    assert (whole == parts).all()


def test_root_steps_report_a_carry_list_overflow(gpu_ctx):
    """With slices of 256 turn-steps the carry list holds two steps' worth of playouts (1 + 256 / slice); standstills (a lone frozen
    Pokemon on either side: every playout runs to the 1,000-turn tie, ~4 slices) pile up three steps' worth: the launch that overflows
    says so (sticky error word), it never loses playouts silently."""
    b, d, p, r = _frozen_standstills(64, 0x5EED0)
    n_roots, reps = 4, 256
    b, d, r = b[:n_roots], d[:n_roots], r[:n_roots]
    rs = _RootSteps(gpu_ctx, b, d, r, _seed_prng(n_roots * reps, 77), reps, 256, max_steps=6000)
    try:
        errs = [rs.step()["err"] for _ in range(8)]
        assert errs[0] == 0 and errs[-1] == 1 and errs == sorted(errs)      # sticky once set
    finally:
        rs.close()
    # ... and a caller that reserves room when `carried` approaches the capacity (what oak_amd.dist.RootSteps.finish does) loses nothing:
    # every playout is credited exactly once, to the step the rule names
    from oak_amd import _lib
    lane = _seed_prng(n_roots * reps, 77)
    ref_lane = lane.copy()
    cnt, s2, ex = O.root_steps_reference(b, d, r, ref_lane, reps, 6, 256, max_steps=6000, threads=8)
    rs = _RootSteps(gpu_ctx, b, d, r, lane, reps, 256, max_steps=6000)
    try:
        cap = C.c_uint32()
        k = 0
        while True:
            rec = rs.step(fresh=k < 6)
            assert rec["err"] == 0 and (rec["count"] == cnt[k]).all() and (rec["sum2"] == s2[k]).all() and rec["turn_steps"] == ex[k], k
            _lib.check(rs.lib.oakgpu_root_steps_capacity(rs.h, C.byref(cap)))
            need = 2 * (rec["carried"] + n_roots * reps)         # what the next launch may carry, x 2 for the shards' imbalance
            if need > cap.value:
                _lib.check(rs.lib.oakgpu_root_steps_reserve(rs.h, 2 * need))
            k += 1
            if k >= 6 and rec["carried"] == 0:
                break
        assert cap.value > 2 * n_roots * reps and int(cnt[:k].sum()) == 6 * n_roots * reps
    finally:
        rs.close()


def test_root_steps_host_class_in_a_torch_process():
    """oak_amd.dist.RootSteps (torch tensors, the exchange hook, pinned host record) through tests/root_steps_check.py in a child
    process -- torch must initialise the GPU before the library does."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "root_steps_check.py")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "root steps ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_full_size_config4_sliced_steps_bit_exact(gpu_ctx):
    """BASELINE configs[3] at full size through the sliced search steps: 256 roots x 4,096 fresh playouts per step (the first 256 lanes of
    config 2), slices of 32 turn-steps, two search steps and the drain.  EVERY step's per-root aggregate (count, 2 x value sum)
    and the turn-steps each launch executes equal the oracle's -- 2.1 M playouts, ~210 M turn-steps, every one credited exactly once --
    and the lane streams afterwards are the oracle's."""
    n_roots, reps, steps, slice_ = 256, 4096, 2, 32
    b, d, p, r = O.make_random_ou_batch(n_roots, seed0=0x0A4B00000000)
    lane = _seed_prng(n_roots * reps, 0xC40000000000)
    ref_lane = lane.copy()
    cnt, s2, ex = O.root_steps_reference(b, d, r, ref_lane, reps, steps, slice_, threads=16)
    rs = _RootSteps(gpu_ctx, b, d, r, lane, reps, slice_)
    try:
        k = 0
        while True:
            rec = rs.step(fresh=k < steps)
            assert rec["err"] == 0
            assert (rec["count"] == cnt[k]).all() and (rec["sum2"] == s2[k]).all(), k
            assert rec["turn_steps"] == ex[k], (k, rec["turn_steps"], ex[k])
            k += 1
            if k >= steps and rec["carried"] == 0:
                break
            assert k < cnt.shape[0]
        assert int(cnt[:k].sum()) == n_roots * reps * steps and int(ex.sum()) > 150_000_000
        assert (rs.lane_streams() == ref_lane).all()
    finally:
        rs.close()
