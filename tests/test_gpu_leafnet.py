"""GPU: fused leaf evaluator (K2 encode+embed, K3 fp32-MFMA main net) vs the numpy oracle.
Tolerance from BASELINE.json north_star: |value - reference| <= 1e-5."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import nn_oracle as NN  # noqa: E402
import oracle_lib as O  # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 1e-5
# hidden = value_hidden = 256 (BASELINE configs[2]'s "3x256 MLP"), written by the reference's torch mirror
# (tests/golden/make_nn_goldens.py); the numpy oracle is held to the mirror's outputs for this file in tests/test_nn_oracle.py
NET256 = os.path.join(ROOT, "tests", "golden", "net_256.battle.net")


def _midgame_states(n, steps, seed0):
    """Random OU battles advanced `steps` random turn-steps on the oracle (statuses, boosts,
    volatiles, durations, fainted slots all appear)."""
    b, d, p, r = O.make_random_ou_batch(n, seed0=seed0)
    out, _ = O.rollout_batch(b, d, r, p, max_steps=steps, threads=4)
    return b, d


@pytest.mark.parametrize("tag", ["default", "tiny", "256"])
def test_value_inference_matches_oracle(gpu_ctx, tag):
    from oak_amd.engine import Network
    path = os.path.join(ROOT, "tests", "golden", "net_%s.battle.net" % tag)
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    assert net.shape()[:3] == (onet.fc0.in_dim, onet.fc0.out_dim, onet.v2.out_dim)
    worst = 0.0
    for steps, seed0 in ((0, 1000), (7, 2000), (40, 3000), (90, 4000)):
        b, d = _midgame_states(192, steps, seed0)
        vals, emb = net.value_inference(b, d, return_embedding=True)
        for i in range(b.shape[0]):
            oe = NN.battle_embedding(onet, b[i], d[i])
            assert np.abs(emb[i] - oe).max() <= 2e-5, (steps, i, np.abs(emb[i] - oe).argmax())
            ov = float(onet.main_value(oe))
            worst = max(worst, abs(float(vals[i]) - ov))
    assert worst <= TOL, worst
    net.close()


def test_config3_net_256(gpu_ctx, tmp_path):
    """BASELINE config 3: 768 -> 256 -> 256 -> 256 -> 1 value path, seeded synthetic weights."""
    from oak_amd.engine import Network
    path = NET256
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b, d = _midgame_states(130, 25, 777)   # ragged: not a multiple of the 64-row tile
    vals = net.value_inference(b, d)
    exp = np.array([float(NN.value_inference(onet, b[i], d[i])) for i in range(b.shape[0])])
    assert np.abs(vals - exp).max() <= TOL
    net.close()


def _main_value_f64(onet, emb):
    """The main net's value path in float64 from a given embedding (fp32 weights as stored): the yardstick both main-net
    kernels are measured against."""
    h = emb.astype(np.float64)
    for layer in (onet.fc0, onet.fc1, onet.v2):
        h = layer.W.astype(np.float64) @ h + layer.b.astype(np.float64)
        h = np.maximum(h, 0.0) if onet.activation == 1 else np.clip(h, 0.0, 1.0)
    y = float(onet.v3.W.astype(np.float64)[0] @ h + np.float64(onet.v3.b[0]))
    return 1.0 / (1.0 + np.exp(-y))


@pytest.mark.parametrize("kind", ["config3", "default", "tiny"])
def test_pair_and_triple_main_nets_are_fp32_results(gpu_ctx, tmp_path, kind):
    """k_mainnet_pair (the default) multiplies fp32 values as scaled fp16 pairs -- three fp16 MFMAs per block, one power-of-two
    scale per layer and per batch row -- and k_mainnet_split as exact sums of three bf16 parts (six bf16 MFMAs); both accumulate
    in fp32 and drop only terms below 2^-24 of a product.  Held here to what that claims: against a FLOAT64 evaluation of the
    same embedding their errors are of the size of k_mainnet_wave's (fp32 MFMA) and far inside the 1e-5 bar; the three kernels
    agree to 1e-6; and the mode switch really switches (the results are not bit-identical everywhere)."""
    from oak_amd.engine import Network
    if kind == "config3":
        path = NET256
    else:
        path = os.path.join(ROOT, "tests", "golden", "net_%s.battle.net" % kind)
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b, d = _midgame_states(700, 30, 4242)          # 5 full 128-row groups + a ragged one
    assert net.main_precision() == ("pair", True)       # the default
    v_pair, emb = net.value_inference(b, d, return_embedding=True)
    assert net.set_main_precision("split") == "pair"
    v_split = net.value_inference(b, d)
    assert net.set_main_precision("fp32") == "split"
    v_fp32 = net.value_inference(b, d)
    assert net.set_main_precision("pair") == "fp32"
    ref = np.array([_main_value_f64(onet, emb[i]) for i in range(b.shape[0])])
    e_pair, e_split, e_fp32 = np.abs(v_pair - ref).max(), np.abs(v_split - ref).max(), np.abs(v_fp32 - ref).max()
    assert e_pair <= 1e-6 and e_split <= 1e-6 and e_fp32 <= 1e-6, (e_pair, e_split, e_fp32)
    assert e_pair <= 4 * e_fp32 + 2e-7 and e_split <= 4 * e_fp32 + 2e-7, (e_pair, e_split, e_fp32)
    assert np.abs(v_split - v_fp32).max() <= 1e-6 and np.abs(v_pair - v_fp32).max() <= 1e-6
    assert ((v_split != v_fp32).any() and (v_pair != v_split).any()) or kind != "config3"
    net.close()


def _rewrite_net(src, dst, edit):
    """Copy a .battle.net file with edit(layer index, bias, W) -> (bias, W) applied to each of its 12 Affine blocks
    (order: pokemon_net 0-1, active_net 2-3, fc0 4, fc1 5, value_fc2 6, value_fc3 7, policy heads 8-11)."""
    import struct
    raw = open(src, "rb").read()
    out, off = [raw[:8]], 8
    for i in range(12):
        n_in, n_out = struct.unpack_from("<II", raw, off)
        off += 8
        b = np.frombuffer(raw, "<f4", n_out, off).copy()
        off += 4 * n_out
        W = np.frombuffer(raw, "<f4", n_out * n_in, off).copy().reshape(n_out, n_in)
        off += 4 * n_out * n_in
        b, W = edit(i, b, W)
        out += [struct.pack("<II", n_in, n_out), b.astype("<f4").tobytes(), W.astype("<f4").tobytes()]
    assert off == len(raw)
    open(dst, "wb").write(b"".join(out))


@pytest.mark.parametrize("case", ["late_columns_large", "late_columns_huge", "one_huge_weight", "zero_rows", "small_rows", "dead_columns"])
def test_fp16_pair_main_net_scales(gpu_ctx, tmp_path, case):
    """k_mainnet_pair's power-of-two scales.  fc0 reads a row 64 columns at a time and sets the row's scale from the FIRST chunk (the
    active's block), lowering it -- and rescaling its sums -- when a later chunk holds a larger value: `late_columns_large` multiplies the
    bench Pokemon's embedding net by 2^12 (its blocks come later in the row) and fc0's columns for those blocks by 2^-12 -- the same
    function, and every row rescales.  `late_columns_huge` does the same with 2^60: beyond what a 5-bit exponent carries, the loader's
    column check refuses the pairs and the network runs on the bf16 triples.  `one_huge_weight` plants a 2^19 in fc1: the other rows'
    columns are dwarfed by it, the column check refuses the pairs.  `zero_rows`: an fc0 bias so negative that every hidden value
    of fc0 is zero -- fc1 sees all-zero rows (the scale's floor).  `small_rows`: every fourth output unit of fc1 and of the bench
    Pokemon's embedding net scaled by 2^-20 (nearly dead units, as training leaves them): every weight ROW carries its own scale, so
    the pairs stay and lose nothing; `dead_columns`: input columns at 2^-16 of the layer's largest weight pass the column check (its
    bound is 2^-18).  Each against a float64 evaluation of the edited net from the kernel's own embedding."""
    from oak_amd.engine import Network
    dst = str(tmp_path / "pair_edge.battle.net")
    base = NN.Net(NET256)
    bench_cols = np.array([s_ * base.side_dim + (1 + base.aod) + q * (1 + base.pod) + 1 + o for s_ in range(2) for q in range(5) for o in range(base.pod)])
    shift = 12 if case == "late_columns_large" else 60

    def edit(i, b, W):
        if i == 1 and case.startswith("late_columns"):
            return b * np.float32(2.0 ** shift), W * np.float32(2.0 ** shift)
        if i == 4 and case.startswith("late_columns"):
            W = W.copy()
            W[:, bench_cols] *= np.float32(2.0 ** -shift)
        if i == 5 and case == "one_huge_weight":
            W = W.copy()
            W[0, 0] = np.float32(2.0 ** 19)
        if i == 4 and case == "zero_rows":
            return b - np.float32(1e6), W
        if i in (4, 5) and case == "dead_columns":        # input units whose weights have decayed to 2^-16 of the others': harmless, the pairs stay
            W = W.copy()
            W[:, 3::17] *= np.float32(2.0 ** -16)
        if i in (1, 5) and case == "small_rows":
            b, W = b.copy(), W.copy()
            b[::4] *= np.float32(2.0 ** -20)
            W[::4] *= np.float32(2.0 ** -20)
        return b, W
    _rewrite_net(NET256, dst, edit)
    net, onet = Network(gpu_ctx, path=dst), NN.Net(dst)
    refused = case in ("one_huge_weight", "late_columns_huge")
    assert net.main_precision() == (("split", True) if refused else ("pair", True))
    net.set_main_precision("pair")                        # not honoured where the loader refused it
    assert net.main_precision()[0] == ("split" if refused else "pair")
    b, d = _midgame_states(300, 30, 999)
    v, emb = net.value_inference(b, d, return_embedding=True)
    if case.startswith("late_columns"):                   # the premise: the row's largest value is not in its first 64 columns
        assert (np.abs(emb[:, 64:]).max(axis=1) > 1000 * np.abs(emb[:, :64]).max(axis=1)).mean() > 0.9
    with np.errstate(over="ignore"):
        ref = np.array([_main_value_f64(onet, emb[i]) for i in range(b.shape[0])])
    assert np.isfinite(v).all() and np.abs(v - ref).max() <= 1e-6, (case, np.abs(v - ref).max())
    if case == "small_rows":                              # the embedding's small rows too, held to THEIR size (2^-20 of the others')
        oemb = np.stack([NN.battle_embedding(onet, b[i], d[i]) for i in range(0, b.shape[0], 7)])
        small = np.array([s_ * base.side_dim + (1 + base.aod) + q * (1 + base.pod) + 1 + o for s_ in range(2) for q in range(5) for o in range(0, base.pod, 4)])
        diff = np.abs(emb[::7] - oemb)
        assert diff.max() <= 2e-5 and diff[:, small].max() <= 2e-5 * 2.0 ** -20, (diff.max(), diff[:, small].max())
        assert np.abs(oemb[:, small]).max() > 0           # (they are not all clipped away)
    net.set_main_precision("fp32")
    assert np.abs(net.value_inference(b, d) - v).max() <= 1e-6
    net.close()


@pytest.mark.parametrize("case", ["down20_up20", "down100_up100", "down120_up120", "down100", "down120", "up100", "tiny_weights"])
def test_bf16_triple_main_net_at_the_edges_of_the_exponent_range(gpu_ctx, tmp_path, case):
    """where the low parts of a bf16 triple flush.  Layers of the 256-wide ReLU net are rescaled by powers of two
    (exact in fp32; ReLU commutes with a positive scale, so a layer scaled down and the next scaled up computes the SAME
    function) and the value is held to a float64 evaluation of the rescaled net from the kernel's own embedding:
      * compensated pairs fc0 x 2^-s, fc1 x 2^+s: s = 20 stays on the bf16 pipe (no weight above 2^20); s = 100 and 120 have
        weights far above 2^20 -- the loader runs such a net on fp32 MFMA, says so, and does not honour a request for the split;
      * one layer scaled down alone (2^-100, 2^-120: every low part flushes, the hidden activations are ~1e-30) stays on the
        bf16 pipe: what is lost is absolute and nothing multiplies it back up;
      * fc0 x 2^+100 alone: fp32 MFMA, values saturate exactly like the float64 evaluation;
      * a net whose fc1 holds weights down to 1e-38 (normal and subnormal fp32) next to ordinary ones: bf16 pipe."""
    from oak_amd.engine import Network
    src = NET256
    dst = str(tmp_path / "edge.battle.net")
    dn = {"down20_up20": -20, "down100_up100": -100, "down120_up120": -120, "down100": -100, "down120": -120, "up100": 100}.get(case, 0)
    up = -dn if "_up" in case else 0

    def edit(i, b, W):
        if i == 4 and dn:
            return b * np.float32(2.0 ** dn), W * np.float32(2.0 ** dn)
        if i == 5 and up:
            return b, W * np.float32(2.0 ** up)
        if i == 5 and case == "tiny_weights":
            W = W.copy()
            W[::3, ::5] *= np.float32(1e-30)     # ~1e-32: normal fp32, every bf16 part below m flushes
            W[1::7, 2::11] = np.float32(1e-40)   # subnormal fp32
        return b, W
    _rewrite_net(src, dst, edit)
    net = Network(gpu_ctx, path=dst)
    onet = NN.Net(dst)
    expect_split = case in ("down20_up20", "down100", "down120", "tiny_weights")
    # the fp16 pairs (the default) are scaled per layer and per batch row, so a layer scaled UP, or down by 2^-20, stays on them; a
    # layer scaled down by 2^-100 is beyond the layer scale's range (2^+-100): its low parts are fp16 subnormals, the loader's row
    # check refuses the pairs and the network runs on the triples -- or on fp32 MFMA where those are refused too
    default = "pair" if case in ("down20_up20", "up100", "tiny_weights") else "split" if expect_split else "fp32"
    assert net.main_precision() == (default, expect_split)
    b, d = _midgame_states(300, 30, 555)
    v_pair, emb = net.value_inference(b, d, return_embedding=True)
    with np.errstate(over="ignore"):
        ref = np.array([_main_value_f64(onet, emb[i]) for i in range(b.shape[0])])
    assert np.isfinite(v_pair).all() and np.abs(v_pair - ref).max() <= 1e-6, (case, np.abs(v_pair - ref).max())
    net.set_main_precision("split")                       # not honoured where the loader refused it
    assert net.main_precision()[0] == ("split" if expect_split else "fp32")
    v = net.value_inference(b, d)
    assert np.isfinite(v).all() and np.abs(v - ref).max() <= 1e-6, (case, np.abs(v - ref).max())
    if expect_split:                                       # and the fp32-MFMA kernel agrees with it
        net.set_main_precision("fp32")
        assert np.abs(net.value_inference(b, d) - v).max() <= 1e-6
    if "_up" in case:                                      # the same function as the unscaled net
        plain = Network(gpu_ctx, path=src)
        assert np.abs(plain.value_inference(b, d) - v).max() <= 2e-6
        plain.close()
    net.close()


@pytest.mark.parametrize("case", ["down20_up20", "down110_up110"])
def test_bf16_triple_embedding_nets_at_the_edges_of_the_exponent_range(gpu_ctx, tmp_path, case):
    """The embedding passes multiply as bf16 triples too (round 4); round-4 advice: an embedding first layer scaled by 2^-s in front of
    a second layer scaled by 2^+s is the same function in fp32 (ReLU commutes with a positive scale), but for s ~ 110 the triples'
    low parts flush in the first layer and the second multiplies the loss back up.  A second-layer weight above 2^20 therefore sends
    the embedding nets through the fp32-MFMA form (k_embed_lds); s = 20 stays on the bf16 pipe.  Either way the embedding equals the
    unscaled net's to 2e-5 and the value to 1e-5 of the oracle."""
    from oak_amd.engine import Network
    dst = str(tmp_path / "edge_emb.battle.net")
    s_ = 20 if case == "down20_up20" else 110

    def edit(i, b, W):
        if i in (0, 2):
            return b * np.float32(2.0 ** -s_), W * np.float32(2.0 ** -s_)
        if i in (1, 3):
            return b, W * np.float32(2.0 ** s_)
        return b, W
    _rewrite_net(NET256, dst, edit)
    net, plain = Network(gpu_ctx, path=dst), Network(gpu_ctx, path=NET256)
    onet = NN.Net(NET256)
    b, d = _midgame_states(200, 30, 777)
    v, emb = net.value_inference(b, d, return_embedding=True)
    pv, pemb = plain.value_inference(b, d, return_embedding=True)
    assert np.isfinite(emb).all() and np.abs(emb - pemb).max() <= 2e-5, (case, np.abs(emb - pemb).max())
    exp = np.array([float(NN.value_inference(onet, b[i], d[i])) for i in range(b.shape[0])])
    assert np.abs(v - exp).max() <= TOL and np.abs(pv - exp).max() <= TOL
    net.close()
    plain.close()


def test_non_finite_parameters_are_refused_by_the_loader(gpu_ctx, tmp_path):
    """oakgpu_net_load refuses a parameter file that holds a NaN or an infinity, naming the layer (the reference would load
    it and propagate NaN through every inference; on the bf16 pipe an infinite weight would split into inf + NaN)."""
    from oak_amd._lib import OakGpuError
    from oak_amd.engine import Network
    src = os.path.join(ROOT, "tests", "golden", "net_default.battle.net")
    for layer, name, bad in ((5, "main_net.fc1", np.inf), (0, "pokemon_net.fc0", np.nan), (7, "main_net.value_fc3", -np.inf), (11, "main_net.p2_policy_fc3", np.nan)):
        dst = str(tmp_path / ("bad%d.battle.net" % layer))

        def edit(i, b, W, layer=layer, bad=bad):
            if i == layer:
                if layer == 7:
                    b = b.copy(); b[0] = bad
                else:
                    W = W.copy(); W[W.shape[0] // 2, W.shape[1] // 3] = bad
            return b, W
        _rewrite_net(src, dst, edit)
        with pytest.raises(OakGpuError, match="non-finite parameter.*" + name.replace(".", r"\.")):
            Network(gpu_ctx, path=dst)


@pytest.mark.parametrize("dims", [
    dict(hidden=96, value_hidden=160),                                        # 3 and 5 output blocks: padded to the 4- and 8-wide kernels
    dict(hidden=32, value_hidden=32, pokemon_out=27, active_out=19),          # one block everywhere, embedding dim 312
    dict(hidden=64, value_hidden=224, pokemon_hidden=64, active_hidden=96, pokemon_out=64, active_out=128),   # narrow hidden layers, widest outputs
    dict(hidden=128, value_hidden=64, pokemon_out=33, active_out=97, pokemon_hidden=100, active_hidden=72),   # ragged widths
    dict(hidden=64, value_hidden=32, pokemon_out=99),                         # party output above 64: the tile kernel takes the party pass
])
def test_layer_widths(gpu_ctx, tmp_path, dims):
    """Every template width of the leaf kernels (1 / 2 / 4 / 8 output blocks, padded blocks, ragged hidden widths) and the
    fall-back of the party pass for outputs above 64, plain and cached, against the numpy oracle."""
    from oak_amd import netfile
    from oak_amd.engine import Network
    path = str(tmp_path / "w.battle.net")
    netfile.write_random_net(path, seed=3, activation=1 + (dims["hidden"] % 64 == 0), **dims)
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b, d = _midgame_states(97, 30, 4242)
    vals = net.value_inference(b, d)
    exp = np.array([float(NN.value_inference(onet, b[i], d[i])) for i in range(b.shape[0])])
    assert np.abs(vals - exp).max() <= TOL
    net.close()


@pytest.mark.parametrize("n", [1, 31, 32, 33, 65, 127, 128, 129, 257])   # wave tiles of 32, workgroup groups of 128
def test_ragged_batch_sizes(gpu_ctx, tmp_path, n):
    """Batches around the kernels' 32-item mini-tiles (1 leaf = 10 party items + 2 actives; 33 leaves = a second main-net tile of one row)."""
    from oak_amd.engine import Network
    path = NET256
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b, d = _midgame_states(n, 40, 900 + n)
    vals = net.value_inference(b, d)
    exp = np.array([float(NN.value_inference(onet, b[i], d[i])) for i in range(n)])
    assert vals.shape == (n,) and np.abs(vals - exp).max() <= TOL
    net.close()


def test_empty_leaf_batch(gpu_ctx):
    from oak_amd.engine import Network
    net = Network(gpu_ctx, path=os.path.join(ROOT, "tests", "golden", "net_default.battle.net"))
    z = lambda *s: np.zeros(s, dtype=np.uint8)
    assert net.value_inference(z(0, 384), z(0, 8)).shape == (0,)
    v, l1, l2 = net.value_policy_inference(z(0, 384), z(0, 8), z(0, 9), z(0), z(0, 9), z(0))
    assert v.shape == (0,) and l1.shape == (0, 9) and l2.shape == (0, 9)
    net.close()


def test_tile_form_of_the_embedding_passes_agrees(gpu_ctx, tmp_path):
    """OAKGPU_EMBED_TILE=1 sends both embedding passes through k_embed_lds (the 64-item tile form, otherwise only taken by
    embedding nets wider than the row kernels allow): a second implementation of the same function, run in a child process."""
    import subprocess
    import sys
    from oak_amd.engine import Network
    path = NET256
    b, d = _midgame_states(257, 35, 31337)
    np.save(str(tmp_path / "b.npy"), b)
    np.save(str(tmp_path / "d.npy"), d)
    net = Network(gpu_ctx, path=path)
    vals = net.value_inference(b, d)
    net.close()
    code = ("import sys, numpy as np; sys.path.insert(0, %r); from oak_amd.engine import Context, Network; c = Context(0); "
            "n = Network(c, path=%r); np.save(%r, n.value_inference(np.load(%r), np.load(%r)))"
            % (ROOT, path, str(tmp_path / "v.npy"), str(tmp_path / "b.npy"), str(tmp_path / "d.npy")))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, OAKGPU_EMBED_TILE="1"), timeout=600)
    tile = np.load(str(tmp_path / "v.npy"))
    assert np.abs(tile - vals).max() <= TOL


def test_full_size_batch_properties(gpu_ctx, tmp_path):
    """BASELINE's full size (65,536 leaves, 768-256-256-256-1): properties that need no oracle pass over the whole batch --
    the evaluation is a pure function of the leaf (a permuted batch gives the permuted values, bit for bit, so no result
    depends on which lane / wave / tile a leaf lands in), repeatable, every value a probability -- and a 2,048-leaf sample
    (every 32nd leaf) against the C oracle."""
    from oak_amd.engine import Network
    n = 65536
    path = NET256
    net = Network(gpu_ctx, path=path)
    b, d, p, r = O.make_random_ou_batch(n, seed0=0xF0115123)
    O.rollout_batch(b, d, r, p, max_steps=25, threads=8)
    v0 = net.value_inference(b, d)
    assert v0.shape == (n,) and np.isfinite(v0).all() and (v0 > 0).all() and (v0 < 1).all()
    assert (net.value_inference(b, d) == v0).all()
    perm = np.random.default_rng(5).permutation(n)
    assert (net.value_inference(b[perm], d[perm]) == v0[perm]).all()
    cnet = O.CNet(path)
    idx = np.arange(0, n, 32)
    exp = cnet.value_inference_batch(np.ascontiguousarray(b[idx]), np.ascontiguousarray(d[idx]), threads=8)
    assert np.abs(v0[idx] - exp).max() <= TOL
    cnet.close()
    net.close()


def test_bad_network_files_raise(gpu_ctx, tmp_path):
    from oak_amd.engine import Network
    from oak_amd._lib import OakGpuError
    with pytest.raises(OakGpuError):
        Network(gpu_ctx, path=str(tmp_path / "missing.battle.net"))
    good = open(os.path.join(ROOT, "tests", "golden", "net_tiny.battle.net"), "rb").read()
    with pytest.raises(OakGpuError):
        Network(gpu_ctx, data=good[:-5])          # truncated
    with pytest.raises(OakGpuError):
        Network(gpu_ctx, data=good + b"\x00")      # trailing byte (network.h:60-63)


def test_config3_rollout_with_leaf_eval_every_turn(gpu_ctx, tmp_path):
    """BASELINE config 3 at test size: after EVERY turn-step run value_inference on the new state.
    GPU: rollout(max_steps=1) + leaf eval per turn; oracle: update + numpy network per turn."""
    from oak_amd.engine import Network
    path = NET256
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    n, turns = 96, 30
    b, d, p, r = O.make_random_ou_batch(n, seed0=0xFACE0000)
    gb, gd, gp, gr = b.copy(), d.copy(), p.copy(), r.copy()
    worst = 0.0
    for t in range(turns):
        got = gpu_ctx.rollout(gb, gd, gr, gp, max_steps=1, return_state=True)
        gb, gd, gp, gr = got["battles"], got["durations"], got["prng"], got["results"]
        vals = net.value_inference(gb, gd)
        out, steps = O.rollout_batch(b, d, r, p, max_steps=1)      # in place on b, d, p
        r = out
        assert (gb == b).all() and (gd == d).all() and (gr == r).all(), t
        for i in range(0, n, 7):
            if int(r[i]) & 15:
                continue
            worst = max(worst, abs(float(vals[i]) - float(NN.value_inference(onet, b[i], d[i]))))
    assert worst <= TOL, worst
    net.close()


@pytest.mark.parametrize("tag", ["default", "tiny"])
def test_value_policy_inference_matches_oracle(gpu_ctx, tag):
    """network.h:102-123: value + the legal choices' logits of both policy heads."""
    from oak_amd.engine import Network
    path = os.path.join(ROOT, "tests", "golden", "net_%s.battle.net" % tag)
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b, d = _midgame_states(150, 12, 8080)
    r = np.array([O.LIB.oracle_result_from_state(O.ptr(b[i])) for i in range(b.shape[0])], dtype=np.uint8)
    c1, n1 = gpu_ctx.choices(b, r, 0)
    c2, n2 = gpu_ctx.choices(b, r, 1)
    vals, l1, l2 = net.value_policy_inference(b, d, c1, n1, c2, n2)
    plain = net.value_inference(b, d)
    assert np.abs(vals - plain).max() == 0.0
    worst = 0.0
    for i in range(b.shape[0]):
        ov, o1, o2 = NN.value_policy_inference(onet, b[i], d[i], c1[i, :n1[i]], c2[i, :n2[i]])
        worst = max(worst, abs(float(vals[i]) - float(ov)))
        assert np.abs(l1[i, :n1[i]] - o1).max() <= 2e-5 and np.abs(l2[i, :n2[i]] - o2).max() <= 2e-5, i
        assert (l1[i, n1[i]:] == 0).all() and (l2[i, n2[i]:] == 0).all()
    assert worst <= TOL
    net.close()


def test_poke_engine_eval_matches_oracle(gpu_ctx):
    """PokeEngine::Eval on the GPU (k_poke_engine) against the numpy restatement, on mid-game states with statuses,
    boosts and volatiles in play: raw scores and sigmoid values."""
    from oak_amd import gamedata
    b, d = _midgame_states(400, 30, 2024)
    root = 12.5
    vals, scores = gpu_ctx.poke_engine_eval(b, root_score=root)
    exp_s = np.array([float(NN.poke_engine_score(b[i], gamedata.MOVES)) for i in range(b.shape[0])])
    exp_v = np.array([float(NN.poke_engine_value(b[i], gamedata.MOVES, root)) for i in range(b.shape[0])])
    assert np.abs(scores - exp_s).max() <= 1e-3 and len(np.unique(np.round(exp_s))) > 50
    assert np.abs(vals - exp_v).max() <= TOL


def test_value_policy_inference_config3_net(gpu_ctx, tmp_path):
    """The policy heads behind the WIDEST main net the kernels take (768 -> 256 -> 256; policy fc2 64): k_policy's LDS
    budget (two activation tiles next to the staged weight chunk) is only exercised at this width."""
    import oracle_lib as O
    from oak_amd.engine import Network
    path = NET256
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b, d = _midgame_states(96, 15, 31337)
    r = np.array([O.LIB.oracle_result_from_state(O.ptr(b[i])) for i in range(b.shape[0])], dtype=np.uint8)
    keep = (r & 15) == 0
    b, d, r = b[keep], d[keep], r[keep]
    c1, n1 = gpu_ctx.choices(b, r, 0)
    c2, n2 = gpu_ctx.choices(b, r, 1)
    vals, l1, l2 = net.value_policy_inference(b, d, c1, n1, c2, n2)
    worst = 0.0
    for i in range(0, b.shape[0], 3):
        ev, e1, e2 = NN.value_policy_inference(onet, b[i], d[i], c1[i, :n1[i]], c2[i, :n2[i]])
        worst = max(worst, abs(float(vals[i]) - float(ev)), float(np.abs(l1[i, :n1[i]] - e1).max()), float(np.abs(l2[i, :n2[i]] - e2).max()))
    assert worst <= 2e-5, worst
    net.close()


@pytest.mark.parametrize("ph", [24, 96, 160, 192])
def test_policy_head_widths(gpu_ctx, tmp_path, ph):
    """k_policy_rows pads the policy-hidden width to 32 / 64 / 128 / 256 and keeps fc3's rows in LDS up to 64-wide heads (from
    global memory beyond): every width class against the oracle.  (Rounds 1-2's kernel refused heads above 160 at hidden 256.)"""
    import oracle_lib as O
    from oak_amd.engine import Network
    path = str(tmp_path / ("p%d.battle.net" % ph))
    NN.write_random_net(path, hidden=256, value_hidden=64, policy_hidden=ph, seed=11)
    net, onet = Network(gpu_ctx, path=path), NN.Net(path)
    b, d = _midgame_states(70, 12, 4242)
    r = np.array([O.LIB.oracle_result_from_state(O.ptr(b[i])) for i in range(b.shape[0])], dtype=np.uint8)
    keep = (r & 15) == 0
    b, d, r = b[keep], d[keep], r[keep]
    c1, n1 = gpu_ctx.choices(b, r, 0)
    c2, n2 = gpu_ctx.choices(b, r, 1)
    # both forms of fc2: bf16 triples (the default, with the main net) and fp32 MFMA (what "fp32" selects for the whole net)
    for mode in ("pair", "split", "fp32"):
        net.set_main_precision(mode)
        assert net.main_precision()[0] == mode
        vals, l1, l2 = net.value_policy_inference(b, d, c1, n1, c2, n2)
        for i in range(0, b.shape[0], 3):
            ev, e1, e2 = NN.value_policy_inference(onet, b[i], d[i], c1[i, :n1[i]], c2[i, :n2[i]])
            assert abs(float(vals[i]) - float(ev)) <= TOL
            assert np.abs(l1[i, :n1[i]] - e1).max() <= 2e-5 and np.abs(l2[i, :n2[i]] - e2).max() <= 2e-5, mode
            assert (l1[i, n1[i]:] == 0).all() and (l2[i, n2[i]:] == 0).all()
    net.close()


def test_policy_heads_wider_than_256_are_refused_at_load(gpu_ctx, tmp_path):
    from oak_amd.engine import Network
    bad = str(tmp_path / "p288.battle.net")
    NN.write_random_net(bad, hidden=256, value_hidden=64, policy_hidden=288, seed=11)
    with pytest.raises(RuntimeError, match="policy"):
        Network(gpu_ctx, path=bad)


def test_two_contexts_share_one_network(gpu_ctx):
    """Workspaces (embeddings, policy activations) belong to the CONTEXT (one per stream), not to the network: two
    contexts evaluating the same loaded network, interleaved and with different batch sizes, each get the oracle's values."""
    from oak_amd.engine import Context, Network
    other = Context(0)
    DEFAULT_NET = os.path.join(ROOT, "tests", "golden", "net_default.battle.net")
    net, onet = Network(gpu_ctx, path=DEFAULT_NET), NN.Net(DEFAULT_NET)
    b1, d1 = _midgame_states(300, 10, 77)
    b2, d2 = _midgame_states(90, 25, 78)
    for _ in range(3):
        import ctypes as C
        v1 = net.value_inference(b1, d1)
        # same network handle through the second context
        v2 = np.zeros(b2.shape[0], dtype=np.float32)
        from oak_amd import _lib
        _lib.check(other.lib.oakgpu_leaf_eval(other.handle, net.handle, b2.ctypes.data_as(C.c_void_p), d2.ctypes.data_as(C.c_void_p),
                                              b2.shape[0], v2.ctypes.data_as(C.c_void_p), None))
        for i in range(0, 300, 37):
            assert abs(float(v1[i]) - float(NN.value_inference(onet, b1[i], d1[i]))) <= TOL
        for i in range(0, b2.shape[0], 11):
            assert abs(float(v2[i]) - float(NN.value_inference(onet, b2[i], d2[i]))) <= TOL
    net.close()
    other.close()


def test_cached_leaf_eval_equals_plain_eval_over_a_resident_batch(gpu_ctx, tmp_path):
    """oakgpu_leaf_eval_cached_dev (party-slot embeddings cached by exact identity tags, the GPU form of PokemonCache): a
    resident batch stepped turn by turn on the device -- values AND embeddings are bit-identical to the uncached evaluator
    every turn; then other battles are loaded into the same lanes WITHOUT resetting the tags, and it is still exact."""
    from hipmem import Dev
    from oak_amd import _lib
    from oak_amd.engine import Network
    path = NET256
    net = Network(gpu_ctx, path=path)
    lib, h = gpu_ctx.lib, gpu_ctx.handle
    n = 3001
    b, d, p, r = O.make_random_ou_batch(n, seed0=0xCAC4E)
    gb, gd, gp, gr = Dev(b), Dev(d), Dev(p), Dev(r)
    steps, vals = Dev(np.zeros(n, np.uint32)), Dev(np.zeros(n, np.float32))
    v_plain, v_cached = Dev(np.zeros(n, np.float32)), Dev(np.zeros(n, np.float32))
    e_plain, e_cached = Dev(np.zeros((n, 768), np.float32)), Dev(np.zeros((n, 768), np.float32), fill=0x7F)   # garbage: every slot must get written
    tags = Dev(np.zeros((n, 10, 6), np.uint32), fill=0xFF)
    for turn in range(45):
        if turn == 30:      # new battles in the same lanes, tags NOT reset: a tag is the slot's whole identity
            b2, d2, p2, r2 = O.make_random_ou_batch(n, seed0=0xBEEF00)
            gb.put(b2); gd.put(d2); gp.put(p2); gr.put(r2)
        _lib.check(lib.oakgpu_rollout_dev(h, gb.p, gd.p, gr.p, gp.p, n, 1, 0, gr.p, steps.p, vals.p, gb.p, gd.p))   # one turn, in place
        _lib.check(lib.oakgpu_leaf_eval_dev(h, net.handle, gb.p, gd.p, n, v_plain.p, e_plain.p))
        _lib.check(lib.oakgpu_leaf_eval_cached_dev(h, net.handle, gb.p, gd.p, n, v_cached.p, e_cached.p, tags.p))
        gpu_ctx.synchronize()
        ep, ec = e_plain.host(), e_cached.host()
        bad = np.nonzero((ep != ec).any(axis=1))[0]
        assert bad.size == 0, (turn, int(bad[0]), np.nonzero(ep[bad[0]] != ec[bad[0]])[0][:8])
        assert (v_plain.host() == v_cached.host()).all(), turn
    # and both agree with the oracle on the final states
    fb, fd = gb.host(), gd.host()
    onet = NN.Net(path)
    vc = v_cached.host()
    for i in range(0, n, 97):
        assert abs(float(vc[i]) - float(NN.value_inference(onet, fb[i], fd[i]))) <= TOL
    for x in (gb, gd, gp, gr, steps, vals, v_plain, v_cached, e_plain, e_cached, tags):
        x.free()
    net.close()


def test_active_move_slots_that_differ_from_the_stored_ones(gpu_ctx):
    """The actives pass loads ONE precombined row per move slot when the active and the stored Pokemon hold the same move there
    (leafnet.hip AR_COMBINED) and falls back to two rows otherwise -- Transform / Mimic in play, or here: patched bytes.  Every
    combination per slot: same move, different move, active PP 0, stored PP 0, empty active slot, Struggle's id (never a
    feature), both sides, mixed with untouched leaves inside the same 32-item mini-tiles (the fallback is a per-k-step,
    wave-uniform branch).  Embedding and value against the oracle."""
    from oak_amd.engine import Network
    path = os.path.join(ROOT, "tests", "golden", "net_default.battle.net")
    net = Network(gpu_ctx, path=path)
    onet = NN.Net(path)
    b, d = _midgame_states(320, 12, 8100)
    b = b.copy()
    rng = np.random.default_rng(99)
    patched = 0
    for i in range(b.shape[0]):
        if i % 3 == 2:
            continue  # untouched neighbours
        for side in range(2):
            base = 184 * side
            slot = int(b[i, base + 176])          # order[0]: the active's party slot (1-based), 0 = none
            if slot == 0 or (i + side) % 4 == 3:
                continue
            act = base + 144 + 24                  # active.moves[4] {id, pp}
            sto = base + 24 * (slot - 1) + 10      # stored.moves[4]
            for k in range(4):
                mode = int(rng.integers(0, 7))
                if mode == 0:
                    continue                                                    # as played
                if mode == 1:
                    b[i, act + 2 * k] = rng.integers(1, 165)                    # a different move in the active slot
                    b[i, act + 2 * k + 1] = max(int(b[i, act + 2 * k + 1]), 1)
                elif mode == 2:
                    b[i, act + 2 * k + 1] = 0                                   # active PP 0, stored PP kept
                elif mode == 3:
                    b[i, sto + 2 * k + 1] = 0                                   # stored PP 0, active PP kept
                elif mode == 4:
                    b[i, act + 2 * k] = 0                                       # empty active slot
                elif mode == 5:
                    b[i, act + 2 * k] = 165                                     # Struggle: no feature (battle.h move slots)
                    b[i, act + 2 * k + 1] = 5
                else:
                    b[i, sto + 2 * k] = rng.integers(1, 165)                    # a different move in the stored slot
                    b[i, sto + 2 * k + 1] = max(int(b[i, sto + 2 * k + 1]), 1)
                patched += 1
    assert patched > 300
    vals, emb = net.value_inference(b, d, return_embedding=True)
    worst = 0.0
    for i in range(b.shape[0]):
        oe = NN.battle_embedding(onet, b[i], d[i])
        assert np.abs(emb[i] - oe).max() <= 2e-5, (i, int(np.abs(emb[i] - oe).argmax()))
        worst = max(worst, abs(float(vals[i]) - float(onet.main_value(oe))))
    assert worst <= TOL, worst
    net.close()
