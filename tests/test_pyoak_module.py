"""The pybind11 face of the boundary (oak_amd/pyoak*.so, names of cpp/src/pyoak.cc:428-716).  Host-side parts run here on
the CPU; search / update need the GPU and are marked so."""
import os

import numpy as np
import pytest

from oak_amd.parse import parse_battle, result_from_state

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mod():
    try:
        from oak_amd import pyoak
    except ImportError:      # a fresh checkout: build the module like __graft_entry__.build() does (host-only C++, no GPU needed)
        import sys
        sys.path.insert(0, ROOT)
        import __graft_entry__ as G
        lib = os.path.join(ROOT, "oak_amd", "liboakgpu.so")
        if not os.path.exists(lib):
            G.build()
        G.build_pyoak(lib, [os.path.join(ROOT, "include", f) for f in sorted(os.listdir(os.path.join(ROOT, "include")))])
        from oak_amd import pyoak
    return pyoak


def test_module_exposes_pyoak_names_and_constants():
    m = _mod()
    for name in ("Heap", "Agent", "Input", "Output", "parse_battle", "battle_string", "format", "update", "search", "solve_matrix", "read_battle_data", "cpp_inference",
                 "value_inference", "value_policy_inference"):
        assert hasattr(m, name), name
    a = m.Agent()
    for field in ("budget", "bandit", "eval", "matrix_ucb", "discrete", "table"):       # pyoak.cc:446-453
        assert hasattr(a, field)
    a.budget, a.bandit, a.eval, a.matrix_ucb, a.discrete, a.table = "8s", "exp3-0.1-0.05", "fp", "100-10-5-1.0", False, False
    assert (a.budget, a.bandit, a.eval) == ("8s", "exp3-0.1-0.05", "fp")
    # pyoak.cc:586-596 / nn/default-hyperparameters.h:10-18
    assert (m.pokemon_in_dim, m.active_in_dim, m.pokemon_hidden_dim, m.pokemon_out_dim) == (198, 427, 128, 59)
    assert (m.active_hidden_dim, m.active_out_dim, m.side_out_dim, m.hidden_dim) == (128, 83, 384, 64)
    assert (m.value_hidden_dim, m.policy_hidden_dim, m.policy_out_dim) == (32, 64, 315)
    o = m.Output()
    assert o.iterations == 0 and o.visit_matrix.shape == (9, 9) and o.p1_nash.shape == (9,)
    h = m.Heap()
    assert h.empty() and h.type() == "std::monostate" and h.nodes() == 0
    assert h.update(0, 0, bytes(16)) is False                                           # Heap::update on monostate (search.cc:31-32)
    assert o.p1_prior.shape == (9,) and not o.p1_prior.any() and o.initial_value == 0.0


def test_parse_battle_and_solve_matrix_and_read_battle_data(tmp_path):
    m = _mod()
    s = "starmie seismictoss 101hp slp3 | snorlax seismictoss 1hp"
    inp = m.parse_battle(s)                                                             # seed default 0x123456
    b, d = parse_battle(s)
    assert inp.battle == b.tobytes() and inp.durations == d.tobytes() and inp.result == result_from_state(b)
    text = m.battle_string(inp)                                                         # pyoak.cc:480-485
    assert text == ("Starmie: 32% (101/323) SLP:4 SeismicToss:32 None:0 None:0 None:0 \n--- --- --- 1 --- --- ---\n"
                    "Snorlax: 1% (1/523) SeismicToss:32 None:0 None:0 None:0 \n")
    assert "Iterations: 0" in m.format(inp, m.Output())
    p1, p2, v = m.solve_matrix(np.array([[0.5, 0.0, 1.0], [1.0, 0.5, 0.0], [0.0, 1.0, 0.5]], dtype=np.float32), 256)
    assert p1.dtype == np.float32 and np.allclose(p1, 1 / 3, atol=1e-6) and np.allclose(p2, 1 / 3, atol=1e-6) and abs(v - 0.5) < 1e-6
    with pytest.raises(RuntimeError, match="Expecting 2d array"):                        # pyoak.cc:396-398
        m.solve_matrix(np.zeros(3, dtype=np.float32), 256)
    from oak_amd.frames import write_frames
    rng = np.random.default_rng(1)
    recs = []
    for k in (3, 0, 7):
        ups = [{"m": 2, "n": 3, "c1": 5, "c2": 9, "iterations": 100 + i, "empirical_value": 0.25, "nash_value": 0.75,
                "p1_empirical": [0.5, 0.5], "p1_nash": [1.0, 0.0], "p2_empirical": [0.2, 0.3, 0.5], "p2_nash": [0.0, 0.0, 1.0]} for i in range(k)]
        recs.append(write_frames(rng.integers(0, 256, 384, dtype=np.uint8), 2, ups))
    path = tmp_path / "0.battle.data"
    path.write_bytes(b"".join(recs))
    got = m.read_battle_data(str(path))                                                  # pyoak.cc:43-71
    assert [(bytes(r), n) for r, n in got] == [(recs[0], 3), (recs[1], 0), (recs[2], 7)]
    with pytest.raises(RuntimeError, match="Failed to open file"):
        m.read_battle_data(str(tmp_path / "missing.battle.data"))


@pytest.mark.gpu
def test_search_and_update_through_the_module():
    m = _mod()
    inp = m.parse_battle("starmie seismictoss 101hp slp5 | snorlax seismictoss 1hp")
    agent = m.Agent()
    agent.budget, agent.bandit, agent.eval = str(1 << 16), "exp3-1.0-0.1", "mc"          # search-test.cc:27-31
    out = m.search(inp, m.Heap(), agent, seed=5)
    assert out.iterations == 1 << 16 and abs(out.empirical_value - 0.5) <= 0.03            # search-test.cc:100-103
    assert out.visit_matrix[0, 0] == 1 << 16 and out.visit_matrix.sum() == 1 << 16 and out.duration_ms > 0
    assert abs(out.p1_nash[0] - 1.0) < 1e-9 and abs(out.nash_value - 0.5) <= 0.04
    # a time budget runs whole batches until it has elapsed
    agent.budget = "50ms"
    out = m.search(inp, m.Heap(), agent)
    assert out.iterations > 0 and out.duration_ms >= 50
    # update() drives the battle like pyoak's: choices from the position, durations carried along
    c1, c2 = m.choices(inp)
    assert len(c1) == 1 and len(c2) == 1
    before = inp.battle
    m.update(inp, c1[0], c2[0])
    assert inp.battle != before
    # the evaluators and the reference's error behaviour (std::runtime_error -> RuntimeError, same texts)
    full = m.parse_battle("starmie surf recover psychic thunderwave | rhydon earthquake rockslide bodyslam substitute")
    for ev, bandit in (("fp", "ucb-2.0"), (os.path.join(ROOT, "tests", "golden", "net_default.battle.net"), "pucb-1.5")):
        agent.budget, agent.eval, agent.bandit = "2048", ev, bandit
        out = m.search(full, m.Heap(), agent, batch=256, seed=1)
        assert out.iterations == 2048 and out.m == 4 and out.n == 4 and abs(out.p1_nash.sum() - 1) < 1e-9
        text = m.format(full, out).split("\n")                                          # MCTS::output_string, util/strings.h:61-152
        assert text[0].startswith("Iterations: 2048, Time: ") and text[3] == "Player 1:" and text[4].split() == ["Surf", "Recover", "Psychic", "ThunderWave"]
        assert text[9].split() == ["Earthquake", "RockSlide", "BodySlam", "Substitute"] and text[14] == "EV Matrix:" and text[15].split() == ["Earthqu", "RockSli", "BodySla", "Substit"]
        assert sum(int(x) for ln in text[23:27] for x in ln.split()[1:] if x != "----") == 2048
    for field, value, text in (("budget", "12parsecs", "Invalid search duration specification"), ("bandit", "thompson-1", "Could not parse bandit string"),
                               ("bandit", "ucb", "Could not parse bandit string"), ("matrix_ucb", "1-2-3", "Could not parse MatrixUCB name"),
                               ("eval", "/nonexistent/x.battle.net", "Cannot open network file")):
        bad = m.Agent()
        bad.budget, bad.bandit = "64", "ucb-1.0"
        setattr(bad, field, value)
        with pytest.raises(RuntimeError, match=text):
            m.search(full, m.Heap(), bad)
    bad = m.Agent()
    bad.budget, bad.bandit, bad.eval = "64", "pucb-1.0", "mc"
    with pytest.raises(RuntimeError, match="Contextual bandit"):                            # search.cc:245-250
        m.search(full, m.Heap(), bad)


@pytest.mark.gpu
def test_heap_resume_priors_and_cpp_inference_through_the_module(tmp_path):
    """pyoak.search(input, heap, agent, output) resumes both the heap's tree and the output (pyoak.cc:575-583 ->
    RuntimeSearch::run -> Search::run, mcts.h:153-155); Heap.update promotes the played child (search.cc:27-52); p1/p2_prior
    are the contextual root priors (mcts.h:196-209); cpp_inference replays a game record like pyoak.cc:331-392."""
    m = _mod()
    net_path = os.path.join(ROOT, "tests", "golden", "net_default.battle.net")
    full = m.parse_battle("starmie surf recover psychic thunderwave | rhydon earthquake rockslide bodyslam substitute")
    agent = m.Agent()
    agent.budget, agent.bandit, agent.eval = str(1 << 14), "ucb-1.0", "mc"
    heap = m.Heap()
    o1 = m.search(full, heap, agent, batch=1024, seed=1)
    assert not heap.empty() and "UCB::JointBandit" in heap.type() and heap.nodes() > 16
    o2 = m.search(full, heap, agent, o1, batch=1024, seed=2)
    assert o1.iterations == 1 << 14 and o2.iterations == 1 << 15 and o2.visit_matrix.sum() == 1 << 15
    assert (o2.visit_matrix >= o1.visit_matrix).all() and o2.duration_ms > o1.duration_ms
    i, j = np.unravel_index(np.argmax(o2.visit_matrix), (9, 9))
    obs = m.update(full, o2.p1_choices[i], o2.p2_choices[j])                             # plays the action; returns the observation
    assert isinstance(obs, bytes) and len(obs) == 16
    kept = heap.update(int(i), int(j), obs)
    assert kept in (True, False) and (heap.nodes() > 0) == kept
    if (full.result & 15) == 0:
        o3 = m.search(full, heap, agent, batch=1024, seed=3)
        assert o3.iterations == 1 << 14
    # contextual priors + the per-position inference entry points
    agent.budget, agent.bandit, agent.eval = "512", "pucb-1.0", net_path
    pos = m.parse_battle("starmie surf recover psychic thunderwave | rhydon earthquake rockslide bodyslam substitute")
    o = m.search(pos, m.Heap(), agent, batch=128, seed=4)
    assert abs(o.p1_prior[:o.m].sum() - 1) < 1e-6 and abs(o.p2_prior[:o.n].sum() - 1) < 1e-6 and not o.p1_prior[o.m:].any()
    e = np.exp(o.p1_logit[:o.m].astype(np.float32))
    assert np.allclose(o.p1_prior[:o.m], e / e.sum(dtype=np.float32), atol=1e-6)
    v, l1, l2 = m.value_policy_inference(pos, net_path)
    assert abs(v - o.initial_value) <= 1e-6 and np.allclose(l1, o.p1_logit[:o.m], atol=1e-6) and np.allclose(l2, o.p2_logit[:o.n], atol=1e-6)
    assert abs(m.value_inference(pos, net_path) - v) <= 1e-6
    # cpp_inference over a self-play record: frame 0 is the record's battle with zero durations (pyoak.cc:349-366)
    from oak_amd.engine import Context, Network
    from oak_amd.frames import selfplay_game, read_frames
    ctx = Context(0)
    teams = np.array([[[143, 34, 156, 0, 0]] + [[0] * 5] * 5, [[121, 94, 86, 105, 0]] + [[0] * 5] * 5], dtype=np.uint8)   # snorlax | starmie
    rec, n_frames, result = selfplay_game(ctx, teams, battle_seed=7, iterations=512, batch=128, evaluator="mc", seed=3)
    out = m.cpp_inference(bytes(rec), net_path)
    assert out["value"].shape == (n_frames,) and out["policy_logit"].shape == (n_frames, 2, 9) and out["policy"].shape == (n_frames, 2, 9)
    game = read_frames(rec)[0]
    battle, ups = game["battle"], game["updates"]
    net = Network(ctx, path=net_path)
    assert abs(out["value"][0] - float(net.value_inference(battle.reshape(1, 384), np.zeros((1, 8), np.uint8))[0])) <= 1e-6
    assert ((out["value"] > 0) & (out["value"] < 1)).all()
    k1, k2 = ups[0]["m"], ups[0]["n"]
    assert abs(out["policy"][0, 0, :k1].sum() - 1) < 1e-5 and abs(out["policy"][0, 1, :k2].sum() - 1) < 1e-5 and not out["policy"][0, 0, k1:].any()
    net.close(); ctx.close()
