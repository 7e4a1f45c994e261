#!/usr/bin/env python3
"""Generate tests/golden/oakside_goldens.json.gz: outputs of the reference's own header-only Oak-side code (feature encoders,
cache key, policy index, hidden-variable resampling, PokeEngine score, turn-0 init) on committed input bytes.

Runs ONLY in the authoring container: it executes oracle/_ref/ref_oakside_dump, a binary compiled (oracle/Makefile, target
`ref`) from the reference's headers where they lie under /root/reference.  Read the header of oracle/ref_oakside_dump.cc for
how that binary is compiled -- against the product's include/pkmn.h, because libpkmn's generated header is absent -- and why
nothing of this repo's engine restatement can reach its outputs (it links no pkmn_* function).

The INPUT states are mid-game positions produced by this repo's CPU oracle (random OU battles advanced 0..150 random
turn-steps): they only have to be well-formed bytes with statuses, boosts, volatiles, durations and fainted slots in them;
what the fixture pins is the reference's function of those bytes.  From ~4,000 candidates a greedy pass keeps the states that
add a not-yet-seen feature index (Pokemon 198, ActivePokemon 427), cache key or policy index, plus a few random ones.
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import oracle_lib as O  # noqa: E402

DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_oakside_dump")


def run_dump(kind, payload):
    with tempfile.NamedTemporaryFile(suffix=".bin") as f:
        f.write(payload)
        f.flush()
        return json.loads(subprocess.check_output([DUMP, kind, f.name]))


def candidates():
    rng = np.random.default_rng(20261004)
    bs, ds = [], []
    for k, steps in enumerate((0, 1, 2, 4, 8, 16, 30, 45, 70, 100, 150)):
        b, d, p, r = O.make_random_ou_batch(384, seed0=0x0A4B00000000 + 100000 * (k + 1))
        if steps:
            O.rollout_batch(b, d, r, p, max_steps=steps, threads=4)
        bs.append(b)
        ds.append(d)
    b = np.concatenate(bs)
    d = np.concatenate(ds)
    seeds = rng.integers(0, 2**63, size=len(b), dtype=np.uint64)
    return b, d, seeds


def features(rec):
    f = set()
    for s, side in enumerate(rec["sides"]):
        if side["active"]:
            f.update(("a", i) for i in side["active"]["e_idx"])
            f.add(("ak", side["active"]["key"]))
        for sl in side["slots"]:
            if sl:
                f.update(("p", i) for i in sl["e_idx"])
                f.add(("pk", sl["key"]))
        f.update(("pol", i) for _, i in side["policy"])
    return f


def main():
    b, d, seeds = candidates()
    payload = b"".join(b[i].tobytes() + d[i].tobytes() + int(seeds[i]).to_bytes(8, "little") for i in range(len(b)))
    recs = run_dump("states", payload)
    assert len(recs) == len(b)
    feats = [features(r) for r in recs]
    seen, keep = set(), []
    order = np.random.default_rng(5).permutation(len(b))
    for i in order:   # greedy: anything that adds a new feature
        if feats[i] - seen:
            seen |= feats[i]
            keep.append(int(i))
    # hidden-variable resampling depends on durations: keep extra states with many non-zero duration fields
    busy = sorted(range(len(b)), key=lambda i: -int(np.count_nonzero(d[i])))[:24]
    keep = sorted(set(keep) | set(busy) | set(int(i) for i in order[:8]))
    states = []
    for i in keep:
        r = recs[i]
        rnd = bytes.fromhex(r.pop("randomized"))
        src = b[i].tobytes()
        r["randomized_diff"] = [[o, rnd[o]] for o in range(384) if rnd[o] != src[o]]
        states.append(dict(battle=src.hex(), durations=d[i].tobytes().hex(), seed=int(seeds[i]), **r))

    # turn-0 init: teams read back out of random OU battles (stored species / move ids), fresh seeds
    tb, _, _, _ = O.make_random_ou_batch(24, seed0=0x0A4B00000000 + 7777)
    teams = []
    payload = b""
    for i in range(len(tb)):
        t = bytearray()
        for s in range(2):
            for k in range(6):
                pk = tb[i][184 * s + 24 * k:184 * s + 24 * k + 24]
                t += bytes([int(pk[21]), int(pk[10]), int(pk[12]), int(pk[14]), int(pk[16])])
        seed = int(seeds[i]) ^ 0x5555
        teams.append((bytes(t), seed))
        payload += bytes(t) + seed.to_bytes(8, "little")
    init = run_dump("teams", payload)

    # `.battle.data` records: games of 1 / 14 / 40 updates with random search outputs (doubles, exact in JSON), plus the
    # truncation edges of compress_probs (0, 1, just below k / 65535 steps)
    import struct
    frng = np.random.default_rng(77)
    games, payload = [], b""
    for gi, count in enumerate((1, 14, 40)):
        battle = b[keep[gi]].tobytes()
        result = int(frng.integers(1, 4))
        ups = []
        for u in range(count):
            m, n = int(frng.integers(1, 10)), int(frng.integers(1, 10))
            dd = lambda k: [float(x) for x in frng.dirichlet(np.ones(k))]
            up = dict(m=m, n=n, c1=int(frng.integers(0, 28)), c2=int(frng.integers(0, 28)), iterations=int(frng.integers(1, 1 << 31)),
                      empirical_value=float(frng.random()), nash_value=float(frng.random()),
                      p1_empirical=dd(m), p1_nash=dd(m), p2_empirical=dd(n), p2_nash=dd(n))
            if u == 0:
                up.update(empirical_value=1.0, nash_value=0.0, p1_empirical=[1.0] + [0.0] * (m - 1), p2_nash=[3 / 65535.0 - 1e-12] + dd(n)[1:])
            ups.append(up)
        games.append(dict(battle=battle.hex(), result=result, updates=ups))
        payload += battle + bytes([result]) + struct.pack("<H", count)
        for up in ups:
            pad = lambda v: list(v) + [0.0] * (9 - len(v))
            payload += struct.pack("<BBBBI", up["m"], up["n"], up["c1"], up["c2"], up["iterations"])
            payload += struct.pack("<38d", up["empirical_value"], up["nash_value"], *pad(up["p1_empirical"]), *pad(up["p1_nash"]),
                                   *pad(up["p2_empirical"]), *pad(up["p2_nash"]))
    records = run_dump("frames", payload)
    for g, r in zip(games, records):
        g["record"] = r
    # name matching of parse_battle's words: every name, every proper prefix, case variants and a few non-names
    from oak_amd import gamedata as GD
    toks = set()
    for nm in GD.SPECIES_NAMES + GD.MOVE_NAMES:
        for k in range(1, len(nm) + 1):
            toks.add(nm[:k].lower())
        toks.add(nm)
        toks.add(nm.upper())
        toks.add(nm + "x")
    toks |= {"par", "psn", "brn", "frz", "slp3", "rst2", "100hp", "50%", "lvl50", "body-slam", "body slam".replace(" ", "_"), "(conf:3)",
             "atk+2", "spc=300", "thunderwavee", "farfetchd"}
    toks = sorted(toks)
    pairs = run_dump("names", ("\n".join(toks) + "\n").encode())
    assert len(pairs) == len(toks)
    names = dict(tokens=toks, species=[a for a, _ in pairs], moves=[b_ for _, b_ in pairs])
    out = dict(
        about="reference Oak-side outputs on committed inputs; generated by tests/golden/make_oakside_goldens.py via "
              "oracle/_ref/ref_oakside_dump (reference headers compiled against include/pkmn.h; no pkmn_* function linked)",
        coverage=dict(active_indices=len({i for k, i in seen if k == "a"}), pokemon_indices=len({i for k, i in seen if k == "p"}),
                      pokemon_keys=len({i for k, i in seen if k == "pk"}), policy_indices=len({i for k, i in seen if k == "pol"}),
                      candidates=len(b)),
        states=states,
        teams=[dict(teams=t.hex(), seed=s, battle=h) for (t, s), h in zip(teams, init)],
        frames=games, names=names)
    import gzip
    path = os.path.join(HERE, "oakside_goldens.json.gz")
    with open(path, "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0) as f:
        f.write(json.dumps(out, separators=(",", ":")).encode())
    print(path, os.path.getsize(path), "bytes;", len(states), "states;", out["coverage"])


if __name__ == "__main__":
    main()
