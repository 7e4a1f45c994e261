"""CPU: the numpy leaf-evaluator oracle against goldens produced by the reference's torch mirror."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nn_oracle as NN  # noqa: E402

G = np.load(os.path.join(ROOT, "tests", "golden", "nn_goldens.npz"))


@pytest.mark.parametrize("tag", ["default", "tiny", "256"])
def test_subnetworks_match_reference_torch(tag):
    net = NN.Net(os.path.join(ROOT, "tests", "golden", "net_%s.battle.net" % tag))
    assert net.activation == (2 if tag == "tiny" else 1)
    if tag == "256":   # BASELINE configs[2]'s net: 768 -> 256 -> 256 -> 256 -> 1
        assert (net.fc0.in_dim, net.fc0.out_dim, net.fc1.out_dim, net.v2.out_dim) == (768, 256, 256, 256)
    for x, y in zip(G[tag + "_xp"], G[tag + "_yp"]):
        nz = np.nonzero(x)[0]
        got = net.embed(net.p0, net.p1, nz, x[nz])
        assert np.abs(got - y).max() < 2e-6
    for x, y in zip(G[tag + "_xa"], G[tag + "_ya"]):
        nz = np.nonzero(x)[0]
        got = net.embed(net.a0, net.a1, nz, x[nz])
        assert np.abs(got - y).max() < 2e-6
    for x, y in zip(G[tag + "_xm"], G[tag + "_ym"]):
        assert abs(float(net.main_value(x)) - float(y[0])) < 1e-6
    for x, l1, l2 in zip(G[tag + "_xm"], G[tag + "_l1"], G[tag + "_l2"]):
        p1, p2 = net.policy_logits(x)
        assert np.abs(p1 - l1).max() < 5e-6 and np.abs(p2 - l2).max() < 5e-6


def test_default_file_size_matches_reference():
    # torch.py default BattleNetwork().write_parameters -> 813,436 bytes
    assert os.path.getsize(os.path.join(ROOT, "tests", "golden", "net_default.battle.net")) == 813436


def test_status_index_static_asserts():
    # cpp/include/encode/battle/battle.h:125-140
    exp = {(0x08, 0): 0, (0x10, 0): 1, (0x20, 0): 2, (0x40, 0): 3, (0x88, 0): 0, (7, 1): 4, (6, 2): 5, (5, 3): 6, (4, 4): 7,
           (3, 5): 8, (2, 6): 9, (1, 7): 10, (0x83, 1): 11, (0x82, 2): 12, (0x81, 3): 13}
    for (st, sl), want in exp.items():
        assert NN.status_index(st, sl) == want


def test_encoder_dims_and_embedding_layout():
    import oracle_lib as O
    net = NN.Net(os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    b, d, p, r = O.make_random_ou_batch(4)
    for i in range(4):
        side = b[i][:184]
        idx, val = NN.encode_active_pokemon(side[0:24], side[144:176], 0)
        assert max(idx) < NN.ACTIVE_POKEMON_IN and len(idx) == len(set(idx))
        emb = NN.battle_embedding(net, b[i], d[i])
        assert emb.shape == (768,) and emb[0] == 1.0 and emb[384] == 1.0   # full-HP leads
        v = NN.value_inference(net, b[i], d[i])
        assert 0.0 < float(v) < 1.0


def test_poke_engine_oracle_hand_computed_cases():
    """PokeEngine::Eval restatement (poke-engine-evaluate.h): values worked out by hand from the header's constants."""
    from oak_amd import gamedata
    from oak_amd.parse import parse_battle
    M = gamedata.MOVES
    # full-health 1v1: each side 100 (hp) + 30 (alive) -> 0; sigmoid(0) = 0.5
    b, d = parse_battle("starmie surf | snorlax bodyslam")
    assert float(NN.poke_engine_score(b, M)) == 0.0 and float(NN.poke_engine_value(b, M, 0.0)) == 0.5
    # a paralysed foe: -25 on its side -> +25 for P1
    b, d = parse_battle("starmie surf | snorlax bodyslam par")
    assert float(NN.poke_engine_score(b, M)) == 25.0
    # burned physical attacker with two physical damaging moves (atk > spc): 2 * -25
    b, d = parse_battle("starmie surf | snorlax bodyslam earthquake amnesia brn")
    assert float(NN.poke_engine_score(b, M)) == 50.0
    # sleeping P1 at 101 / 323 hp: 100 * 101 / 323 - 25 + 30 against 130
    b, d = parse_battle("starmie seismictoss 101hp slp3 | snorlax seismictoss")
    exp = np.float32(np.float32(np.float32(100) * np.float32(101)) / np.float32(323)) - np.float32(25) + np.float32(30) - np.float32(130)
    assert abs(float(NN.poke_engine_score(b, M)) - float(exp)) < 1e-4
    assert abs(float(NN.poke_engine_value(b, M, 0.0)) - 1 / (1 + np.exp(-0.0125 * float(exp)))) < 1e-6


def test_c_port_matches_numpy_oracle(tmp_path):
    """oracle/nn_host.c (bench.py's CPU baseline for leaf-evals/s) against nn_oracle.py -- which the torch-mirror goldens
    pin -- on mid-game states: embeddings and values, default (relu), tiny (clamp) and the 3x256 config-3 network."""
    import oracle_lib as O
    b, d, p, r = O.make_random_ou_batch(160, seed0=0xC0DE)
    O.rollout_batch(b[:80], d[:80], r[:80], p[:80], max_steps=25, threads=2)
    O.rollout_batch(b[80:], d[80:], r[80:], p[80:], max_steps=70, threads=2)
    wide = os.path.join(ROOT, "tests", "golden", "net_256.battle.net")
    for path in (os.path.join(ROOT, "tests", "golden", "net_default.battle.net"),
                 os.path.join(ROOT, "tests", "golden", "net_tiny.battle.net"), wide):
        net, cnet = NN.Net(path), O.CNet(path)
        vals = cnet.value_inference_batch(b, d, threads=3)
        for i in range(b.shape[0]):
            e = NN.battle_embedding(net, b[i], d[i])
            ce = cnet.embedding(b[i], d[i], dim=e.shape[0])
            assert np.abs(e - ce).max() <= 2e-6, (path, i, int(np.abs(e - ce).argmax()))
            assert abs(float(net.main_value(e)) - float(vals[i])) <= 1e-5
        cnet.close()
