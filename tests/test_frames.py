"""CPU: the `.battle.data` record codec (oakgpu_frames_write / _read, host code) against a record assembled by hand from
the layout of Train::Battle::CompressedFrames (cpp/include/train/battle/compressed-frame.h:37-243)."""
import struct

import numpy as np
import pytest

from oak_amd.frames import read_frames, write_frames


def _hand_record(battle, result, updates):
    body = b""
    for u in updates:
        q = lambda x: int(x * 65535.0)                       # compress_probs<double, uint16_t> (:11-25): truncation
        body += struct.pack("<BBBIHH", (u["m"] - 1) | ((u["n"] - 1) << 4), u["c1"], u["c2"], u["iterations"], q(u["empirical_value"]),
                            q(u["nash_value"]))
        for name in ("p1_empirical", "p1_nash", "p2_empirical", "p2_nash"):
            body += b"".join(struct.pack("<H", q(x)) for x in u[name])
    total = 4 + 2 + 384 + 1 + len(body)
    return struct.pack("<IH", total, len(updates)) + bytes(battle) + bytes([result]) + body


def _updates(rng, count):
    ups = []
    for _ in range(count):
        m, n = int(rng.integers(1, 10)), int(rng.integers(1, 10))
        d = lambda k: rng.dirichlet(np.ones(k))
        ups.append({"m": m, "n": n, "c1": int(rng.integers(0, 28)), "c2": int(rng.integers(0, 28)), "iterations": int(rng.integers(1, 1 << 30)),
                    "empirical_value": float(rng.random()), "nash_value": float(rng.random()), "p1_empirical": d(m), "p1_nash": d(m),
                    "p2_empirical": d(n), "p2_nash": d(n)})
    return ups


def test_record_layout_matches_the_reference_format_byte_for_byte():
    rng = np.random.default_rng(5)
    battle = rng.integers(0, 256, 384, dtype=np.uint8)
    ups = _updates(rng, 37)
    ups[0].update(m=1, n=1, p1_empirical=[1.0], p1_nash=[1.0], p2_empirical=[1.0], p2_nash=[0.0], empirical_value=1.0, nash_value=0.0)
    rec = write_frames(battle, 0x02, ups)
    assert rec == _hand_record(battle, 0x02, ups)
    assert len(rec) == 391 + sum(1 + 2 + 4 + 4 + 4 * (u["m"] + u["n"]) for u in ups)   # Update::n_bytes_static (:77-82), max 83


def test_read_back_and_concatenated_records():
    rng = np.random.default_rng(6)
    games = [(rng.integers(0, 256, 384, dtype=np.uint8), int(rng.integers(1, 4)), _updates(rng, k)) for k in (1, 12, 0, 80)]
    blob = b"".join(write_frames(b, r, u) for b, r, u in games)
    back = read_frames(blob)
    assert len(back) == 4
    for (b, r, ups), g in zip(games, back):
        assert (g["battle"] == b).all() and g["result"] == r and len(g["updates"]) == len(ups)
        for u, v in zip(ups, g["updates"]):
            assert (u["m"], u["n"], u["c1"], u["c2"], u["iterations"]) == (v["m"], v["n"], v["c1"], v["c2"], v["iterations"])
            assert abs(u["empirical_value"] - v["empirical_value"]) <= 1 / 65535 and abs(u["nash_value"] - v["nash_value"]) <= 1 / 65535
            for name in ("p1_empirical", "p1_nash", "p2_empirical", "p2_nash"):
                assert np.abs(np.asarray(u[name]) - v[name]).max() <= 1 / 65535      # uncompress_probs (:27-35)
        # writing what was read is the identity on bytes
    assert b"".join(write_frames(g["battle"], g["result"], g["updates"]) for g in back) == blob


def test_malformed_records_are_refused():
    rng = np.random.default_rng(7)
    rec = write_frames(rng.integers(0, 256, 384, dtype=np.uint8), 1, _updates(rng, 5))
    for bad in (rec[:100], rec[:-3], struct.pack("<I", len(rec) + 50) + rec[4:], rec[:4] + struct.pack("<H", 6) + rec[6:]):
        with pytest.raises(RuntimeError):
            read_frames(bad)
    with pytest.raises(RuntimeError):
        write_frames(np.zeros(384, np.uint8), 1, [{"m": 10, "n": 1, "c1": 0, "c2": 0, "iterations": 1, "empirical_value": 0.5, "nash_value": 0.5,
                                                     "p1_empirical": [0.1] * 9, "p1_nash": [0.1] * 9, "p2_empirical": [1.0], "p2_nash": [1.0]}])


def test_endless_battle_check_follows_the_generators_rule():
    """oakgpu_endless_battle_check = generate.cc:127-152: every pairing Ghost against Ghost, neither with a move that can hit a Ghost
    (type Normal / Fighting or base power 0).  The library's move mask is recomputed here from the data file; the 16 sample teams never
    trigger it (which is why the TUTORIAL's frames-per-game statistic cannot depend on it)."""
    import ctypes as C
    import json
    import os
    from oak_amd import _lib, gamedata as G
    lib = _lib.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = json.load(open(os.path.join(root, "oak_amd", "data", "gen1_data.json")))
    cant = {0} | {i + 1 for i, m in enumerate(data["moves"]) if m[2] in (0, 1) or m[1] == 0}

    def check(teams):
        t = np.ascontiguousarray(np.array(teams, dtype=np.uint8).reshape(60))
        return lib.oakgpu_endless_battle_check(t.ctypes.data_as(C.c_void_p))

    def rule(teams):
        t = np.array(teams).reshape(2, 6, 5)
        ghost = lambda s: s[0] in (92, 93, 94)
        blunt = lambda s: all(int(m) in cant for m in s[1:])
        return int(all(ghost(a) and blunt(b) and ghost(b) and blunt(a) for a in t[0] for b in t[1]))
    rng = np.random.default_rng(7)
    gengar_blunt = [94, G.match_move("hypnosis"), G.match_move("bodyslam"), G.match_move("confuseray"), G.match_move("explosion")]
    gengar_sharp = [94, G.match_move("hypnosis"), G.match_move("nightshade"), G.match_move("thunderbolt"), G.match_move("explosion")]
    assert check([[gengar_blunt] * 6, [gengar_blunt] * 6]) == 1
    assert check([[gengar_blunt] * 6, [gengar_blunt] * 5 + [gengar_sharp]]) == 0
    assert check([[gengar_blunt] * 5 + [[0, 0, 0, 0, 0]], [gengar_blunt] * 6]) == 0        # an empty slot is not a Ghost
    for _ in range(300):                                                                    # every move id against the mask
        teams = [[[int(rng.choice([92, 93, 94, 113])), *[int(x) for x in rng.integers(0, 166, 4)]] for _ in range(6)] for _ in range(2)]
        if rng.random() < 0.7:
            teams = [[[94, *[int(rng.choice(sorted(cant))) for _ in range(4)]] for _ in range(6)] for _ in range(2)]
            if rng.random() < 0.5:
                teams[int(rng.integers(0, 2))][int(rng.integers(0, 6))][int(rng.integers(1, 5))] = int(rng.integers(0, 166))
        assert check(teams) == rule(teams)
    sample = json.load(open(os.path.join(root, "tests", "golden", "ou_sample_teams.json")))["teams"]
    tb = [[[G.match_species(s[0])] + [G.match_move(m) for m in s[1:]] for s in t] for t in sample]
    assert not any(check([a, b]) for a in tb for b in tb)
