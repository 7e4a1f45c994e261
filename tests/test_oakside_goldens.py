"""The oracle's Oak-side pieces (feature encoders, policy index, hidden-variable resampling, PokeEngine score, turn-0 init)
against outputs of the REFERENCE's own header-only code on the same bytes (tests/golden/oakside_goldens.json.gz, made by
tests/golden/make_oakside_goldens.py through oracle/_ref/ref_oakside_dump -- see that source's header for how it is built).

These pin SURVEY 8 rows a7 / a8 / a11 / a13 / a14 (encoders) / f3 (policy index) / f4 (PokeEngine) mechanically; they say
nothing about the libpkmn boundary (a1 / a2), which stays unpinned."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nn_oracle as NN  # noqa: E402
import oracle_lib as O  # noqa: E402

import gzip  # noqa: E402

with gzip.open(os.path.join(HERE, "golden", "oakside_goldens.json.gz"), "rt") as f:
    G = json.load(f)
STATES = G["states"]


def _bytes(h):
    return np.frombuffer(bytes.fromhex(h), dtype=np.uint8).copy()


def _dur(d, s):
    return int.from_bytes(bytes(d[4 * s:4 * s + 4]), "little")


def test_fixture_covers_the_feature_space():
    c = G["coverage"]
    assert len(STATES) >= 150 and c["active_indices"] >= 400 and c["pokemon_indices"] >= 185 and c["policy_indices"] >= 300
    assert sum(1 for s in STATES for side in s["sides"] if side["active"] is None) > 0          # a fainted lead
    assert sum(1 for s in STATES for side in s["sides"] for sl in side["slots"] if sl is None) > 0


def test_party_slot_encoder_matches_the_reference():
    """Encode::Battle::Pokemon::write, sparse (battle.h:208-214), indices AND order AND values, + the hp fraction of
    network.h:164."""
    n = 0
    for st in STATES:
        b, d = _bytes(st["battle"]), _bytes(st["durations"])
        for s in range(2):
            side = b[184 * s:184 * (s + 1)]
            dur = _dur(d, s)
            for slot in range(2, 7):
                ref = st["sides"][s]["slots"][slot - 2]
                pid = int(side[176 + slot - 1])
                pk = side[24 * (pid - 1):24 * pid] if pid else None
                alive = pid != 0 and NN._u16(pk, 18) != 0
                assert alive == (ref is not None)
                if not alive:
                    continue
                idx, val = NN.encode_pokemon(pk, (dur >> (3 * (slot - 1))) & 7)
                assert idx == ref["e_idx"]
                assert np.array_equal(np.array(val, dtype=np.float32), np.array(ref["e_val"], dtype=np.float32))
                assert np.float32(NN._u16(pk, 18)) / np.float32(NN._u16(pk, 0)) == np.float32(ref["hp"])
                n += 1
    assert n > 1000


def test_active_encoder_matches_the_reference():
    """Encode::Battle::ActivePokemon::write, sparse (battle.h:544-551): stats, types, boosts, volatiles, move slots, durations,
    then the stored Pokemon."""
    n = 0
    for st in STATES:
        b, d = _bytes(st["battle"]), _bytes(st["durations"])
        for s in range(2):
            side = b[184 * s:184 * (s + 1)]
            ref = st["sides"][s]["active"]
            sid = int(side[176]) - 1
            stored = side[24 * sid:24 * sid + 24]
            assert (NN._u16(stored, 18) != 0) == (ref is not None)
            if ref is None:
                continue
            idx, val = NN.encode_active_pokemon(stored, side[144:176], _dur(d, s))
            assert idx == ref["e_idx"]
            assert np.array_equal(np.array(val, dtype=np.float32), np.array(ref["e_val"], dtype=np.float32))
            assert np.float32(NN._u16(stored, 18)) / np.float32(NN._u16(stored, 0)) == np.float32(ref["hp"])
            n += 1
    assert n > 350


def test_policy_index_matches_the_reference():
    """Encode::Battle::Policy::get_index (policy.h:29-58) for every well-formed move / switch choice byte."""
    for st in STATES:
        b = _bytes(st["battle"])
        for s in range(2):
            for c, want in st["sides"][s]["policy"]:
                assert NN.policy_index(b[184 * s:184 * (s + 1)], c) == want


def test_hidden_variable_resampling_matches_the_reference():
    """MCTS::randomize_hidden_variables (durations.h:25-97) after battle.rng = seed (mcts.h:255-257), every byte."""
    changed = 0
    for st in STATES:
        b, d = _bytes(st["battle"]), _bytes(st["durations"])
        want = b.copy()
        for o, v in st["randomized_diff"]:
            want[o] = v
        b[376:384] = np.frombuffer(int(st["seed"]).to_bytes(8, "little"), dtype=np.uint8)
        O.LIB.oracle_randomize_hidden_variables(O.ptr(b), O.ptr(d))
        assert np.array_equal(b, want)
        changed += any(o < 376 for o, _ in st["randomized_diff"])
    assert changed > 40   # the fixture does exercise the resampling, not only the seed copy


def test_poke_engine_score_matches_the_reference():
    """PokeEngine::evaluate_battle and Eval::evaluate at the state's own root score (poke-engine-evaluate.h:184-204)."""
    from oak_amd import gamedata
    for st in STATES:
        b = _bytes(st["battle"])
        got = float(NN.poke_engine_score(b, gamedata.MOVES))
        assert abs(got - st["pe_score"]) <= 1e-4 * max(1.0, abs(st["pe_score"]))
        assert st["pe_value_at_root"] == 0.5 and float(NN.poke_engine_value(b, gamedata.MOVES, got)) == 0.5


def test_result_from_state_matches_the_reference():
    """PKMN::result(battle) (pkmn.h:235-272): the request byte recomputed from a state -- oracle and the host-side mirror."""
    from oak_amd.parse import result_from_state
    kinds = set()
    for st in STATES:
        b = _bytes(st["battle"])
        assert int(O.LIB.oracle_result_from_state(O.ptr(b))) == st["result"] == int(result_from_state(b))
        kinds.add(st["result"])
    assert len(kinds) >= 3   # plain turns and forced switches on either side at least


def test_name_matching_of_battle_strings_matches_the_reference():
    """PKMN::string_to_species / string_to_move (libpkmn/strings.h:53-83,313-331): unique case-insensitive prefix, and an
    ambiguous prefix is NOT rescued by an exact name ("mew", "thunder")."""
    from oak_amd import gamedata as GD
    N = G["names"]
    assert len(N["tokens"]) > 2000
    for tok, sp, mv in zip(N["tokens"], N["species"], N["moves"]):
        assert GD._unique_prefix(GD.SPECIES_NAMES, 12, tok) == sp, tok
        assert GD._unique_prefix(GD.MOVE_NAMES, 13, tok) == mv, tok
    look = dict(zip(N["tokens"], zip(N["species"], N["moves"])))
    assert look["mew"][0] == -1 and look["thunder"][1] == -1 and look["mewt"][0] == 150 and look["thunderw"][1] == 86
    with pytest.raises(RuntimeError, match="Could not match string to Species"):
        from oak_amd.parse import parse_battle
        parse_battle("mew psychic | snorlax bodyslam")


def test_battle_string_matches_the_reference():
    """PKMN::battle_data_to_string (libpkmn/strings.h:187-303) = pyoak.battle_string, character for character."""
    from oak_amd.parse import battle_string
    kinds = 0
    for st in STATES:
        got = battle_string(_bytes(st["battle"]), _bytes(st["durations"]))
        assert got == st["text"]
        kinds += ">>" in got
    assert kinds > 20


def test_turn0_init_matches_the_reference():
    """PKMN::battle / Init::init_side / init_pokemon / compute_stat (pkmn.h:50-57, init.h:90-154), all 384 bytes."""
    for t in G["teams"]:
        got = O.init_battle(_bytes(t["teams"]).reshape(2, 6, 5), int(t["seed"]))
        assert got.tobytes().hex() == t["battle"]


def test_battle_data_records_match_the_reference_writer():
    """oakgpu_frames_write (host code of the product library) against Train::Battle::CompressedFrames::write of the reference
    on the same search outputs (compressed-frame.h:84-118,181-214), every byte; and read back."""
    from oak_amd.frames import read_frames, write_frames
    assert len(G["frames"]) == 3
    for g in G["frames"]:
        rec = write_frames(_bytes(g["battle"]), g["result"], g["updates"])
        assert rec.hex() == g["record"]
        back = read_frames(bytes.fromhex(g["record"]))
        assert len(back) == 1 and len(back[0]["updates"]) == len(g["updates"]) and back[0]["result"] == g["result"]


def test_c_port_embedding_follows_the_reference_encoders():
    """The plain-C port (oracle/nn_host.c) builds its embedding from its own encoders: rebuild the embedding from the
    REFERENCE's sparse lists and the numpy layers, and compare."""
    path = os.path.join(HERE, "golden", "net_default.battle.net")
    net = NN.Net(path)
    cnet = O.CNet(path)
    for st in STATES[::4]:
        b, d = _bytes(st["battle"]), _bytes(st["durations"])
        want = _reference_embedding(net, st)
        got = cnet.embedding(b, d)
        assert np.abs(got - want).max() <= 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# the HIP path against the same reference outputs (through the C ABI)
# ---------------------------------------------------------------------------------------------------------------------
def _reference_embedding(net, st):
    want = np.zeros(2 * net.side_dim, dtype=np.float32)
    for s in range(2):
        base = s * net.side_dim
        a = st["sides"][s]["active"]
        if a:
            want[base] = np.float32(a["hp"])
            want[base + 1:base + 1 + net.aod] = net.embed(net.a0, net.a1, a["e_idx"], [np.float32(v) for v in a["e_val"]])
        for k, sl in enumerate(st["sides"][s]["slots"]):
            if sl:
                o = base + (1 + net.aod) + k * (1 + net.pod)
                want[o] = np.float32(sl["hp"])
                want[o + 1:o + 1 + net.pod] = net.embed(net.p0, net.p1, sl["e_idx"], [np.float32(v) for v in sl["e_val"]])
    return want


@pytest.mark.gpu
def test_gpu_embedding_follows_the_reference_encoders(gpu_ctx):
    """k_embed_prows / k_embed_arows encode on the device; the 768-float embedding they produce must be the one the
    REFERENCE's sparse feature lists give through the same two layers (numpy, fp32)."""
    from oak_amd.engine import Network
    path = os.path.join(HERE, "golden", "net_default.battle.net")
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b = np.stack([_bytes(st["battle"]) for st in STATES])
    d = np.stack([_bytes(st["durations"]) for st in STATES])
    vals, emb = net.value_inference(b, d, return_embedding=True)
    for i, st in enumerate(STATES):
        want = _reference_embedding(onet, st)
        assert np.abs(emb[i] - want).max() <= 1e-5, i
        assert abs(float(vals[i]) - float(onet.main_value(want))) <= 1e-5, i
    net.close()


@pytest.mark.gpu
def test_gpu_policy_logits_sit_at_the_reference_indices(gpu_ctx):
    """k_policy gathers logit[get_index(side, choice)]: compare with the numpy heads read at the REFERENCE's indices."""
    from oak_amd.engine import Network
    path = os.path.join(HERE, "golden", "net_default.battle.net")
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    n = len(STATES)
    b = np.stack([_bytes(st["battle"]) for st in STATES])
    d = np.stack([_bytes(st["durations"]) for st in STATES])
    ch = [np.zeros((n, 9), dtype=np.uint8) for _ in range(2)]
    cnt = [np.zeros(n, dtype=np.uint8) for _ in range(2)]
    for i, st in enumerate(STATES):
        for s in range(2):
            pol = st["sides"][s]["policy"]
            cnt[s][i] = len(pol)
            ch[s][i, :len(pol)] = [c for c, _ in pol]
    vals, l1, l2 = net.value_policy_inference(b, d, ch[0], cnt[0], ch[1], cnt[1])
    for i, st in enumerate(STATES):
        h1, h2 = onet.policy_logits(_reference_embedding(onet, st))
        for s, (got, head) in enumerate(((l1, h1), (l2, h2))):
            want = np.array([head[k] for _, k in st["sides"][s]["policy"]], dtype=np.float32)
            assert np.abs(got[i, :len(want)] - want).max() <= 2e-5, (i, s)
    net.close()


@pytest.mark.gpu
def test_gpu_turn0_init_matches_the_reference(gpu_ctx):
    """k_init against PKMN::battle(p1, p2, seed) of the reference, all 384 bytes."""
    teams = np.stack([_bytes(t["teams"]) for t in G["teams"]]).reshape(-1, 2, 6, 5)
    seeds = np.array([int(t["seed"]) for t in G["teams"]], dtype=np.uint64)
    b, _, _ = gpu_ctx.battle(teams, seeds, first_update=False)
    for i, t in enumerate(G["teams"]):
        assert b[i].tobytes().hex() == t["battle"], i


@pytest.mark.gpu
def test_gpu_poke_engine_matches_the_reference(gpu_ctx):
    """k_poke_engine's raw score against PokeEngine::evaluate_battle of the reference."""
    b = np.stack([_bytes(st["battle"]) for st in STATES])
    vals, scores = gpu_ctx.poke_engine_eval(b, root_score=0.0)
    for i, st in enumerate(STATES):
        assert abs(float(scores[i]) - st["pe_score"]) <= 1e-4 * max(1.0, abs(st["pe_score"])), i
        assert abs(float(vals[i]) - 1.0 / (1.0 + np.exp(-0.0125 * st["pe_score"]))) <= 1e-5, i


@pytest.mark.gpu
def test_gpu_hidden_variable_resampling_matches_the_reference(gpu_ctx):
    """The prep of the rollout kernels (battle.rng = device draw; randomize_hidden_variables) against the reference's output:
    one playout of zero turn-steps from the state, the device draw being the fixture's seed."""
    changed = 0
    busy = [st for st in STATES if any(o < 376 for o, _ in st["randomized_diff"])]
    for st in busy + STATES[:16]:
        b, d = _bytes(st["battle"]), _bytes(st["durations"])
        want = b.copy()
        for o, v in st["randomized_diff"]:
            want[o] = v
        res = int(O.LIB.oracle_result_from_state(O.ptr(b)))
        out = gpu_ctx.rollout_shared_device(b, d, res, np.array([st["seed"], 0, 0, 0], dtype=np.uint64), 1, max_steps=0,
                                            prep=True, return_state=True)
        assert np.array_equal(out["battles"][0], want)
        changed += any(o < 376 for o, _ in st["randomized_diff"])
    assert changed > 40
