"""CPU: pin the oracle's Oak-side code against reference-generated known answers."""
import ctypes as C
import json
import os

import numpy as np

import oracle_lib as O
from oak_amd import gamedata as G

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KA = json.load(open(os.path.join(ROOT, "tests", "golden", "rng_known_answers.json")))

# PKMN::battle(benchmark_teams[0], benchmark_teams[1], 1111111) before the first update
APPENDIX_B_BATTLE = (
    "4d01c600a800200120013b088e105e109c104d01007ccd64bf026c006c00c600"
    "34013a102f1887105518bf02007100642f012001ca01ee000c013b0880109908"
    "3f082f01005bd9649d0166015201b200bc00221859109d10a4109d0100705464"
    "4301f8000c0148012a013b0869205518562043010079c96461012a0120013e01"
    "ee003b08221859103f0861010080006400000000000000000000000000000000"
    "0000000000000000000000000000000001020304050600003901c600bc005201"
    "70015e1069204520562039010041cc64bf026c006c00c6003401732045208710"
    "5620bf0200710064890120010c01d0005c0199085e104f184e3089010067ca64"
    "cf010c010201da0020013b083f082f185518cf010083d9640b023e01e4009e00"
    "e400221859103f0878080b02008f006461012a0120013e01ee003b0822185910"
    "3f08610100800064000000000000000000000000000000000000000000000000"
    "00000000000000000102030405060000000000000000000047f4100000000000")

BENCHMARK_TEAMS = [  # cpp/include/teams/benchmark-teams.h:15-28
    [("Jynx", ["Blizzard", "LovelyKiss", "Psychic", "Rest"]), ("Chansey", ["IceBeam", "Sing", "SoftBoiled", "Thunderbolt"]),
     ("Cloyster", ["Blizzard", "Clamp", "Explosion", "HyperBeam"]), ("Rhydon", ["BodySlam", "Earthquake", "RockSlide", "Substitute"]),
     ("Starmie", ["Blizzard", "Recover", "Thunderbolt", "ThunderWave"]), ("Tauros", ["Blizzard", "BodySlam", "Earthquake", "HyperBeam"])],
    [("Alakazam", ["Psychic", "Recover", "SeismicToss", "ThunderWave"]), ("Chansey", ["Reflect", "SeismicToss", "SoftBoiled", "ThunderWave"]),
     ("Exeggutor", ["Explosion", "Psychic", "SleepPowder", "StunSpore"]), ("Lapras", ["Blizzard", "HyperBeam", "Sing", "Thunderbolt"]),
     ("Snorlax", ["BodySlam", "Earthquake", "HyperBeam", "SelfDestruct"]), ("Tauros", ["Blizzard", "BodySlam", "Earthquake", "HyperBeam"])],
]


def benchmark_teams():
    return [[[G.species_id(s)] + [G.move_id(m) for m in ms] for s, ms in team] for team in BENCHMARK_TEAMS]


def test_fast_prng_seed_and_stream():
    for seed, v in KA["fast_prng"].items():
        st = np.zeros(8, dtype=np.uint8)
        O.LIB.oracle_fast_prng_seed(O.ptr(st), C.c_uint64(int(seed)))
        assert list(st) == v["state"]
        got = [O.LIB.oracle_fast_prng_uniform_64(O.ptr(st)) for _ in range(16)]
        assert [str(x) for x in got] == v["uniform_64"]
        assert [O.LIB.oracle_fast_prng_next32(O.ptr(st)) % 9 for _ in range(8)] == v["random_int_9"]


def test_mt19937_uniform_64():
    for seed, v in KA["mt19937_uniform_64"].items():
        st = np.zeros(626 * 4, dtype=np.uint8)
        O.LIB.oracle_mt19937_seed(O.ptr(st), int(seed))
        assert [str(O.LIB.oracle_mt19937_uniform_64(O.ptr(st))) for _ in range(16)] == v


def test_engine_lcg_stream():
    # PKMN::RNG::next, cpp/include/libpkmn/rng.h:9-11 (the oracle advances battle.rng with it)
    for seed, v in KA["lcg"].items():
        s = int(seed)
        for want in v:
            s = (0x5D588B656C078965 * s + 0x269EC3) & (2**64 - 1)
            assert str(s) == want
    assert "%016x" % int(KA["lcg"]["1111111"][0]) == "fec21490bb81fdc6"


def test_init_battle_matches_appendix_b():
    b = O.init_battle(benchmark_teams(), 1111111)
    assert b.tobytes().hex() == APPENDIX_B_BATTLE
    assert O.LIB.oracle_result_from_state(O.ptr(b)) == 0x50
    opt = O.Options()
    assert O.update(b, 0, 0, opt) == 0x50 and b[368] == 1     # leads sent out, turn 1


def test_config1_benchmark_playouts_are_reproducible():
    """BASELINE config 1: 2 fixed teams, 1k playouts driven by a shared mt19937{1111111}
    (benchmark.cc:23-31 + mcts.h:250-263,448-496).  Pins the oracle against itself across builds:
    the (steps, result, state-hash) digest is committed in tests/golden/config1_digest.json."""
    teams = benchmark_teams()
    b0 = O.init_battle(teams, 1111111)
    opt = O.Options()
    res0 = O.update(b0, 0, 0, opt)
    dev = np.zeros(626 * 4, dtype=np.uint8)
    O.LIB.oracle_mt19937_seed(O.ptr(dev), 1111111)
    total, wins, digest = 0, 0, 0xcbf29ce484222325
    for _ in range(1000):
        b = b0.copy()
        d = np.zeros(8, dtype=np.uint8)
        r = np.array([O.LIB.oracle_mt19937_uniform_64(O.ptr(dev))], dtype=np.uint64)
        b[376:384] = r.view(np.uint8)
        O.LIB.oracle_randomize_hidden_variables(O.ptr(b), O.ptr(d))
        steps = C.c_uint32(0)
        res = O.LIB.oracle_rollout_mt(O.ptr(b), O.ptr(d), res0, O.ptr(dev), 100000, C.byref(steps))
        total += steps.value
        wins += (res & 15) == 1
        digest = (digest ^ O.LIB.oracle_hash64(O.ptr(b), 384)) * 0x100000001b3 % 2**64
    path = os.path.join(ROOT, "tests", "golden", "config1_digest.json")
    got = {"turn_steps": total, "p1_wins": int(wins), "digest": "%016x" % digest}
    if not os.path.exists(path):
        json.dump(got, open(path, "w"))
    assert got == json.load(open(path))


def test_randomize_hidden_variables_ranges():
    """cpp/include/search/durations.h:25-97: hidden counters drawn from the public durations."""
    from oak_amd.parse import parse_battle
    for c in range(1, 6):
        b, d = parse_battle("starmie seismictoss (conf:%d) | snorlax bodyslam" % c)
        seen = set()
        for seed in range(200):
            bb = b.copy()
            bb[376:384] = np.array([seed * 0x9E3779B97F4A7C15 % 2**64], dtype=np.uint64).view(np.uint8)
            O.LIB.oracle_randomize_hidden_variables(O.ptr(bb), O.ptr(d))
            vol = int.from_bytes(bytes(bb[160:168]), "little")
            seen.add((vol >> 18) & 7)
        lo, hi = (2, 5) if c == 1 else (1, 6 - c)
        assert seen == set(range(lo, hi + 1)), (c, seen)
    for k in range(0, 7):
        b, d = parse_battle("starmie seismictoss slp%d | snorlax bodyslam" % k)
        seen = set()
        for seed in range(300):
            bb = b.copy()
            bb[376:384] = np.array([seed * 0x9E3779B97F4A7C15 % 2**64], dtype=np.uint64).view(np.uint8)
            O.LIB.oracle_randomize_hidden_variables(O.ptr(bb), O.ptr(d))
            seen.add(int(bb[20]) & 7)
        assert seen == set(range(1, 8 - (k + 1) + 1)), (k, seen)
