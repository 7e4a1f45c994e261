"""CPU: the product library loads and exports every symbol include/oakgpu.h declares; the
host-side mirror imports; no compute call is made (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from oak_amd import _lib
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "oakgpu.h")).read()
    declared = set(re.findall(r"\b(oakgpu_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for name in declared:
        assert getattr(lib, name) is not None, name
    pk = open(os.path.join(ROOT, "include", "pkmn.h")).read()
    pk_declared = set(re.findall(r"\b(pkmn_(?:gen1_battle|result)_[a-z0-9_]+)\s*\(", pk))
    assert pk_declared == set(_lib.PKMN_SYMBOLS), pk_declared ^ set(_lib.PKMN_SYMBOLS)
    for name in pk_declared:
        assert getattr(lib, name) is not None, name


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        return
    from oak_amd import _lib
    from oak_amd.engine import Context
    try:
        Context(0)
    except _lib.OakGpuError:
        return
    raise AssertionError("Context() must fail without a HIP device (no CPU fallback)")


def test_product_never_references_the_oracle():
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "oak_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".inc", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle_lib|liboracle|nn_oracle|oracle/", txt):
                    bad.append(f)
    assert not bad, bad


def test_cpp_host_layer_compiles_and_links(tmp_path):
    """include/oakgpu.hpp + include/pkmn.h compile as C++17 and link against liboakgpu.so."""
    import subprocess
    exe = str(tmp_path / "cpp_host_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp_host_smoke.cc"), "-L", os.path.join(ROOT, "oak_amd"), "-loakgpu",
                           "-Wl,-rpath," + os.path.join(ROOT, "oak_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, OAKGPU_SMOKE_NET=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_host_layer_runs_on_the_gpu_and_per_leaf_eval_equals_batched(tmp_path):
    """The GPU half of tests/cpp_host_smoke.cc: a rollout through OakGPU::BatchedMonteCarlo, the refusals of TreeSearch / run,
    and the reference's per-leaf eval signatures -- OakGPU::Network::value_inference(battle, durations) and
    value_policy_inference(b, d, m, n, c1, c2, p1, p2) (nn/battle/network.h:72-79,102-123, as mcts.h:196-209,401-422 calls them)
    -- against the batched calls on the same leaves: identical values and logits."""
    import subprocess
    exe = str(tmp_path / "cpp_host_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp_host_smoke.cc"), "-L", os.path.join(ROOT, "oak_amd"), "-loakgpu",
                           "-Wl,-rpath," + os.path.join(ROOT, "oak_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    env = dict(os.environ, OAKGPU_SMOKE_NET=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "per-leaf eval == batched eval" in out.stdout and out.stdout.strip().endswith("ok"), out.stdout


def test_host_mt19937_fill_matches_reference_known_answers():
    """oakgpu_mt19937_fill (host side of the shared-generator rollouts) against the std::mt19937 / uniform_64 values
    dumped from the reference's util/random.h (tests/golden/rng_known_answers.json).  No GPU needed."""
    import json
    from oak_amd.engine import mt19937_uniform_64
    ka = json.load(open(os.path.join(ROOT, "tests", "golden", "rng_known_answers.json")))["mt19937_uniform_64"]
    for seed, want in ka.items():
        assert [str(int(x)) for x in mt19937_uniform_64(int(seed), len(want))] == want
        assert [str(int(x)) for x in mt19937_uniform_64(int(seed), 4, skip=3)] == want[3:7]
