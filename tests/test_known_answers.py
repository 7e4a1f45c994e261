"""CPU: the reference's only runtime tests (cpp/src/search-test.cc:50-109) on the oracle.
Every position has exactly one legal joint action, so the search value is the mean playout value."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from oak_amd.parse import parse_battle, result_from_state

POSITIONS = [("starmie seismictoss 1hp (conf:5) | snorlax bodyslam 1hp", 1.0, 0.0),
             ("starmie seismictoss 1hp (conf:4) | snorlax bodyslam 1hp", .5 + .5 / 2, .03),
             ("starmie seismictoss 1hp (conf:3) | snorlax bodyslam 1hp", .33 + .66 / 2, .03),
             ("starmie seismictoss 1hp (conf:2) | snorlax bodyslam 1hp", .25 + .75 / 2, .03),
             ("starmie seismictoss 1hp (conf:1) | snorlax bodyslam 1hp", .5, .03),
             ("starmie seismictoss 1hp slp6 | snorlax seismictoss 1hp", 0.0, 0.0)]
POSITIONS += [("starmie seismictoss 101hp slp%d | snorlax seismictoss 1hp" % k, 1.0 / (7 - k), 0.0 if k == 6 else .03)
              for k in range(7)]


@pytest.mark.parametrize("pos,expected,err", POSITIONS)
def test_search_test_position(pos, expected, err):
    n = 20000
    b, d = parse_battle(pos, 4321)
    res = result_from_state(b)
    for s in (0, 1):   # one legal joint action
        assert len(O.choices(b, s, (res >> (4 + 2 * s)) & 3)) == 1
    prng = np.zeros((n, 8), dtype=np.uint8)
    for i in range(n):
        O.LIB.oracle_fast_prng_seed(O.ptr(prng[i]), C.c_uint64(31337 + i))
    out, _ = O.rollout_batch(np.tile(b, (n, 1)), np.tile(d, (n, 1)), np.full(n, res, np.uint8), prng, prep=True, threads=4)
    t = out & 15
    value = float(((t == 1) + 0.5 * (t == 3)).mean())
    assert abs(value - expected) <= err + 1e-9, (pos, value)
