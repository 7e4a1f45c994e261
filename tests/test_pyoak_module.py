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
    for name in ("Heap", "Agent", "Input", "Output", "parse_battle", "update", "search", "solve_matrix", "read_battle_data"):
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
    assert m.Heap().empty()


def test_parse_battle_and_solve_matrix_and_read_battle_data(tmp_path):
    m = _mod()
    s = "starmie seismictoss 101hp slp3 | snorlax seismictoss 1hp"
    inp = m.parse_battle(s)                                                             # seed default 0x123456
    b, d = parse_battle(s)
    assert inp.battle == b.tobytes() and inp.durations == d.tobytes() and inp.result == result_from_state(b)
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
