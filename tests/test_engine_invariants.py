"""CPU: structural invariants of the oracle engine on random OU playouts (SURVEY 8c ii):
HP <= max, PP never increases, result byte consistent with PKMN::result(battle), determinism,
and legal-choice enumeration edge cases."""
import numpy as np

import oracle_lib as O
from oak_amd.parse import parse_battle, result_from_state


def _hp(b):
    pk = b[:, :368].reshape(-1, 2, 184)[:, :, :144].reshape(-1, 2, 6, 24)
    return pk[..., 18].astype(int) + 256 * pk[..., 19].astype(int), pk[..., 0].astype(int) + 256 * pk[..., 1].astype(int)


def test_random_ou_playouts_invariants():
    n = 3000
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x77770000)
    out, steps = O.rollout_batch(b, d, r, p, max_steps=1000, threads=4)
    hp, mx = _hp(b)
    assert (hp <= mx).all()
    t = out & 15
    assert ((t >= 0) & (t <= 3)).all()          # never ERROR
    for i in range(n):
        if t[i] in (1, 2):
            assert (O.LIB.oracle_result_from_state(O.ptr(b[i])) & 15) == t[i]
    turn = b[:, 368].astype(int) + 256 * b[:, 369].astype(int)
    assert (turn[t == 0] < 1000).all() and (steps[t == 0] == 1000).all()   # only the step cap leaves NONE
    assert 60 < steps.mean() < 160


def test_stepwise_monotone_pp_and_request_consistency():
    n = 400
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x12340000)
    rng = np.random.default_rng(3)
    opts = [O.Options(d[i]) for i in range(n)]
    for step in range(80):
        for i in range(n):
            if int(r[i]) & 15:
                continue
            want = O.LIB.oracle_result_from_state(O.ptr(b[i]))
            assert want == r[i], (step, i, hex(want), hex(int(r[i])))     # pkmn.h:235-272
            c1s = O.choices(b[i], 0, (int(r[i]) >> 4) & 3)
            c2s = O.choices(b[i], 1, (int(r[i]) >> 6) & 3)
            assert 1 <= len(c1s) <= 9 and 1 <= len(c2s) <= 9
            opts[i].set()
            r[i] = O.update(b[i], int(c1s[rng.integers(len(c1s))]), int(c2s[rng.integers(len(c2s))]), opts[i])


def test_determinism_same_seed_same_bytes():
    a = O.make_random_ou_batch(256, seed0=42)
    b = O.make_random_ou_batch(256, seed0=42)
    ra = O.rollout_batch(a[0], a[1], a[3], a[2])
    rb = O.rollout_batch(b[0], b[1], b[3], b[2])
    assert (a[0] == b[0]).all() and (ra[0] == rb[0]).all() and (ra[1] == rb[1]).all()


def test_choices_edge_cases():
    # forced switch after a faint; a side whose last Pokemon fainted; Struggle when out of PP
    b, d = parse_battle("starmie surf 0hp; snorlax bodyslam | alakazam psychic")
    res = result_from_state(b)
    assert res == (2 << 4)                              # p1 must switch, p2 passes
    assert list(O.choices(b, 0, 2)) == [(2 << 2) | 2] and list(O.choices(b, 1, 0)) == [0]
    b, d = parse_battle("starmie surf:0 | alakazam psychic")
    assert list(O.choices(b, 0, 1)) == [1]              # move with data 0 = Struggle
    b, d = parse_battle("starmie surf 0hp | alakazam psychic")
    assert result_from_state(b) == 2                    # p1 has nothing left: LOSE
