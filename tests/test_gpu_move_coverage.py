"""GPU parity, move by move: for every gen-1 move (1..164) a Pokemon that knows it uses it whenever it is a
legal choice, against random opponents; every intermediate battle / durations / actions / result byte of the
batched GPU update must equal the oracle's.  Random OU play exercises common moves millions of times but
rare effects (Transform, Mimic, Metronome, Bide, Counter, Conversion, Haze, ...) only occasionally."""
import numpy as np
import pytest

import oracle_lib as O
from oak_amd import gamedata as G

pytestmark = pytest.mark.gpu


def _teams():
    legal, pools, sizes = G.ou_pools()
    rng = np.random.default_rng(2024)
    users = [G.species_id(s) for s in ("Mew", "Snorlax", "Gengar", "Starmie", "Rhydon", "Ditto", "Jolteon", "Chansey")]
    support = [G.move_id(m) for m in ("Tackle", "Recover", "Substitute", "ThunderWave", "Toxic", "Rest", "Agility", "Surf")]
    teams, forced = [], []
    for move in range(1, 165):
        for rep in range(6):
            t = np.zeros((2, 6, 5), np.uint8)
            for s in range(2):
                sp = rng.choice(legal, 6, replace=False)
                for k in range(6):
                    t[s, k, 0] = sp[k]
                    m = min(4, sizes[sp[k]])
                    t[s, k, 1:1 + m] = rng.choice(pools[sp[k], :sizes[sp[k]]], m, replace=False)
            side = rep % 2                                   # the move's user alternates between P1 and P2
            t[side, 0, 0] = users[(move + rep) % len(users)]
            others = [x for x in rng.permutation(support) if x != move][:3]
            t[side, 0, 1:5] = [move] + others
            teams.append(t)
            forced.append((side, move))
    return np.array(teams), forced


def test_every_move_stepwise_bit_exact(gpu_ctx):
    teams, forced = _teams()
    n = teams.shape[0]
    seeds = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(12345))
    gb, gd, gr = gpu_ctx.battle(teams, seeds, first_update=True)
    b = np.stack([O.init_battle(teams[i], int(seeds[i])) for i in range(n)])
    opts = [O.Options() for _ in range(n)]
    r = np.array([O.update(b[i], 0, 0, opts[i]) for i in range(n)], dtype=np.uint8)
    d = np.stack([o.durations for o in opts])
    assert (gb == b).all() and (gr == r).all() and (gd == d).all()
    rng = np.random.default_rng(99)
    used = np.zeros(166, dtype=np.int64)
    for step in range(36):
        c1 = np.zeros(n, np.uint8)
        c2 = np.zeros(n, np.uint8)
        for i in range(n):
            if int(r[i]) & 15:
                continue
            side, move = forced[i]
            picks = []
            for pl in (0, 1):
                ch = O.choices(b[i], pl, (int(r[i]) >> (4 + 2 * pl)) & 3)
                pick = int(ch[rng.integers(len(ch))])
                if pl == side:
                    for c in ch:                           # prefer the move under test when it is selectable
                        if (int(c) & 3) == 1 and (int(c) >> 2) >= 1 and b[i][184 * pl + 144 + 24 + 2 * ((int(c) >> 2) - 1)] == move:
                            pick = int(c)
                            used[move] += 1
                picks.append(pick)
            c1[i], c2[i] = picks
        live = (r & 15) == 0
        gres, gact = gpu_ctx.update(gb, c1, c2, gd)
        for i in np.nonzero(live)[0]:
            opts[i].set()
            r[i] = O.update(b[i], int(c1[i]), int(c2[i]), opts[i])
            d[i] = opts[i].durations
        acts = np.stack([o.actions for o in opts])
        bad = np.nonzero(live & ((gb != b).any(axis=1) | (gres != r) | (gd != d).any(axis=1) | (gact != acts).any(axis=1)))[0]
        assert bad.size == 0, "step %d lane %d move %s" % (step, bad[0], G.MOVE_NAMES[forced[bad[0]][1]])
        gb[~live] = b[~live]
        gd[~live] = d[~live]
    assert (used[1:165] > 0).all(), [G.MOVE_NAMES[m] for m in range(1, 165) if used[m] == 0]


def test_every_move_rollout_bit_exact(gpu_ctx):
    """Same battles through the register-resident rollout kernel (different engine implementation)."""
    teams, _ = _teams()
    n = teams.shape[0]
    seeds = (np.arange(n, dtype=np.uint64) * np.uint64(0xD1B54A32D192ED03) + np.uint64(777))
    gb, gd, gr = gpu_ctx.battle(teams, seeds, first_update=True)
    import ctypes as C
    prng = np.zeros((n, 8), dtype=np.uint8)
    for i in range(n):
        O.LIB.oracle_fast_prng_seed(O.ptr(prng[i]), C.c_uint64(555 + i))
    got = gpu_ctx.rollout(gb, gd, gr, prng, max_steps=1000, return_state=True)
    ob, od, op = gb.copy(), gd.copy(), prng.copy()
    oout, osteps = O.rollout_batch(ob, od, gr, op, max_steps=1000, threads=8)
    assert (got["steps"] == osteps).all() and (got["results"] == oout).all()
    assert (got["battles"] == ob).all() and (got["durations"] == od).all() and (got["prng"] == op).all()


def test_every_move_through_the_tree_step_bit_exact(gpu_ctx):
    """The same move-by-move walks through oakgpu_tree_step_dev -- since round 5 the REGISTER-resident engine with its own chance-action
    tracking (k_tree_step_staged): every battle / durations / result / action-key byte and both players' next choices against the
    oracle, every move used at least once, with a damage-roll clamp on every third level."""
    from hipmem import Dev
    teams, forced = _teams()
    n = teams.shape[0]
    seeds = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(4242))
    b = np.stack([O.init_battle(teams[i], int(seeds[i])) for i in range(n)])
    opts = [O.Options() for _ in range(n)]
    r = np.array([O.update(b[i], 0, 0, opts[i]) for i in range(n)], dtype=np.uint8)
    d = np.stack([o.durations for o in opts])
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    db, dd, dr = Dev(b), Dev(d), Dev(r)
    dc1, dc2, dact = Dev(np.zeros(n, np.uint8)), Dev(np.zeros(n, np.uint8)), Dev(np.zeros((n, 16), np.uint8))
    dch1, dch2, dn1, dn2 = Dev(np.zeros((n, 9), np.uint8)), Dev(np.zeros((n, 9), np.uint8)), Dev(np.zeros(n, np.uint8)), Dev(np.zeros(n, np.uint8))
    rng = np.random.default_rng(199)
    used = np.zeros(166, dtype=np.int64)
    roll_byte = lambda rolls, seed: 236 if rolls == 1 else 217 + (38 // (rolls - 1)) * (seed % rolls)
    for step in range(36):
        rolls = 3 if step % 3 == 2 else 39
        c1, c2 = np.full(n, 0xFF, np.uint8), np.full(n, 0xFF, np.uint8)
        live = (r & 15) == 0
        for i in np.nonzero(live)[0]:
            side, move = forced[i]
            picks = []
            for pl in (0, 1):
                ch = O.choices(b[i], pl, (int(r[i]) >> (4 + 2 * pl)) & 3)
                pick = int(ch[rng.integers(len(ch))])
                if pl == side:
                    for c in ch:
                        if (int(c) & 3) == 1 and (int(c) >> 2) >= 1 and b[i][184 * pl + 144 + 24 + 2 * ((int(c) >> 2) - 1)] == move:
                            pick = int(c)
                            used[move] += 1
                picks.append(pick)
            c1[i], c2[i] = picks
        dc1.put(c1)
        dc2.put(c2)
        assert lib.oakgpu_tree_step_dev(h, db.p, dd.p, dr.p, dc1.p, dc2.p, n, rolls, dact.p, dch1.p, dn1.p, dch2.p, dn2.p) == 0
        gpu_ctx.synchronize()
        gb, gd, gr, gact, gch1, gch2, gn1, gn2 = db.host(), dd.host(), dr.host(), dact.host(), dch1.host(), dch2.host(), dn1.host(), dn2.host()
        for i in np.nonzero(live)[0]:
            over = np.zeros(16, np.uint8)
            if rolls != 39:
                over[0], over[8] = roll_byte(rolls, int(b[i][376 + 6])), roll_byte(rolls, int(b[i][376 + 7]))
            opts[i].set(None, over if rolls != 39 else None)
            r[i] = O.update(b[i], int(c1[i]), int(c2[i]), opts[i])
            d[i] = opts[i].durations
            name = G.MOVE_NAMES[forced[i][1]]
            assert (gact[i] == opts[i].actions).all(), (step, i, name, "actions")
            if (int(r[i]) & 15) == 0:
                o1, o2 = O.choices(b[i], 0, (int(r[i]) >> 4) & 3), O.choices(b[i], 1, (int(r[i]) >> 6) & 3)
                assert gn1[i] == len(o1) and (gch1[i, :len(o1)] == o1).all() and gn2[i] == len(o2) and (gch2[i, :len(o2)] == o2).all(), (step, i, name)
        bad = np.nonzero((gb != b).any(axis=1) | (gr != r) | (gd != d).any(axis=1))[0]
        assert bad.size == 0, "step %d lane %d move %s" % (step, bad[0], G.MOVE_NAMES[forced[bad[0]][1]])
    assert (used[1:165] > 0).all(), [G.MOVE_NAMES[m] for m in range(1, 165) if used[m] == 0]
    for x in (db, dd, dr, dc1, dc2, dact, dch1, dch2, dn1, dn2):
        x.free()
