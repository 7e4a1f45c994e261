"""One search level at a time through oakgpu_tree_step_dev against the oracle: every battle byte, durations, result, the 16-byte
chance-action key and both players' legal choices, over random walks with finished lanes (0xFF) and all five damage-roll clamps.
Run in-process by tests/test_gpu_parity.py (the default kernel: the register-resident engine, staged) and as a script in a child
process with OAKGPU_TREE_STEP=lds (the LDS-resident engine's kernel) -- the switch is read once per process.
usage: python tests/tree_step_check.py [n] [levels]   (exit code 0 = every byte equal)"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402  (the checker)
from hipmem import Dev  # noqa: E402


def roll_byte(rolls, seed):   # mcts.h:569-604
    return 236 if rolls == 1 else 217 + (38 // (rolls - 1)) * (seed % rolls)


def run(ctx, n=1500, levels=40, seed0=0x7EE50000):
    lib, h = ctx.lib, ctx.handle
    b, d, p, r = O.make_random_ou_batch(n, seed0=seed0)
    opts = [O.Options(d[i]) for i in range(n)]
    rng = np.random.default_rng(17)
    db, dd, dr = Dev(b), Dev(d), Dev(r)
    dc1, dc2 = Dev(np.zeros(n, np.uint8)), Dev(np.zeros(n, np.uint8))
    dact = Dev(np.zeros((n, 16), np.uint8))
    dch1, dch2 = Dev(np.zeros((n, 9), np.uint8)), Dev(np.zeros((n, 9), np.uint8))
    dn1, dn2 = Dev(np.zeros(n, np.uint8)), Dev(np.zeros(n, np.uint8))
    done = np.zeros(n, bool)
    P = lambda x: x.p
    checked = 0
    for level in range(levels):
        rolls = (39, 3, 20, 1, 2)[level % 5]
        c1, c2 = np.full(n, 0xFF, np.uint8), np.full(n, 0xFF, np.uint8)
        live = ((r & 15) == 0) & ~done & (rng.random(n) > 0.05)   # 5 % of the running lanes sit a level out: 0xFF, untouched
        for i in np.nonzero(live)[0]:
            o1 = O.choices(b[i], 0, (int(r[i]) >> 4) & 3)
            o2 = O.choices(b[i], 1, (int(r[i]) >> 6) & 3)
            c1[i], c2[i] = o1[rng.integers(len(o1))], o2[rng.integers(len(o2))]
        dc1.put(c1)
        dc2.put(c2)
        sentinel = np.full((n, 16), 0xA5, np.uint8)
        dact.put(sentinel)
        rc = lib.oakgpu_tree_step_dev(h, P(db), P(dd), P(dr), P(dc1), P(dc2), n, rolls, P(dact), P(dch1), P(dn1), P(dch2), P(dn2))
        assert rc == 0, rc
        ctx.synchronize()
        gb, gd, gr, gact = db.host(), dd.host(), dr.host(), dact.host()
        gch1, gch2, gn1, gn2 = dch1.host(), dch2.host(), dn1.host(), dn2.host()
        for i in np.nonzero(live)[0]:
            over = np.zeros(16, np.uint8)
            if rolls != 39:
                over[0], over[8] = roll_byte(rolls, int(b[i][376 + 6])), roll_byte(rolls, int(b[i][376 + 7]))
            opts[i].set(None, over if rolls != 39 else None)
            r[i] = O.update(b[i], int(c1[i]), int(c2[i]), opts[i])
            d[i] = opts[i].durations
            assert (gact[i] == opts[i].actions).all(), (level, i, "actions")
            if (int(r[i]) & 15) == 0:
                o1 = O.choices(b[i], 0, (int(r[i]) >> 4) & 3)
                o2 = O.choices(b[i], 1, (int(r[i]) >> 6) & 3)
                assert gn1[i] == len(o1) and (gch1[i, :len(o1)] == o1).all() and (gch1[i, len(o1):] == 0).all(), (level, i, "p1 choices")
                assert gn2[i] == len(o2) and (gch2[i, :len(o2)] == o2).all() and (gch2[i, len(o2):] == 0).all(), (level, i, "p2 choices")
            else:
                assert gn1[i] == 0 and gn2[i] == 0, (level, i, "terminal counts")
            checked += 1
        assert (gact[~live] == 0xA5).all(), (level, "a finished lane's action key was written")
        assert (gr == r).all(), (level, "results", int(np.nonzero(gr != r)[0][0]))
        bad = np.nonzero((gb != b).any(axis=1))[0]
        assert bad.size == 0, (level, "battle", int(bad[0]), bool(live[bad[0]]))
        assert (gd == d).all(), (level, "durations")
        done |= rng.random(n) < 0.02    # a few descents end at every level
    for x in (db, dd, dr, dc1, dc2, dact, dch1, dch2, dn1, dn2):
        x.free()
    return checked


if __name__ == "__main__":
    from oak_amd.engine import Context
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    levels = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    print("tree step levels checked against the oracle:", run(Context(0), n, levels), "lane-levels, kernel:", os.environ.get("OAKGPU_TREE_STEP", "staged"))
