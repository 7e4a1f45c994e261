"""`.battle.net` parameter files (host utility).

Format (reference reader cpp/include/nn/affine.h:35-70, nn/battle/network.h:52-70, header
cpp/src/search.cc:127-131; reference writer src/oak/torch.py:135-138,388-392): 8-byte header whose byte 0 is
`activation - 1` (0 relu, 1 clamp), then 12 Affine blocks `u32 in, u32 out, f32 bias[out], f32 W[out][in]` in
the order pokemon_net(2) active_net(2) fc0 fc1 value_fc2 value_fc3 p1_policy_fc2/3 p2_policy_fc2/3.
"""
import struct

import numpy as np

POKEMON_IN, ACTIVE_IN, POLICY_OUT = 198, 427, 315


def layer_dims(pokemon_hidden=128, pokemon_out=59, active_hidden=128, active_out=83, hidden=64, value_hidden=32,
               policy_hidden=64):
    side = (1 + active_out) + 5 * (1 + pokemon_out)
    return [(POKEMON_IN, pokemon_hidden), (pokemon_hidden, pokemon_out), (ACTIVE_IN, active_hidden),
            (active_hidden, active_out), (2 * side, hidden), (hidden, hidden), (hidden, value_hidden), (value_hidden, 1),
            (hidden, policy_hidden), (policy_hidden, POLICY_OUT), (hidden, policy_hidden), (policy_hidden, POLICY_OUT)]


def write_random_net(path, seed=7, activation=1, **dims):
    """Seeded synthetic network with the reference's initialiser U(-1/sqrt(in), 1/sqrt(in))
    (Affine::initialize, affine.h:105-115).  BASELINE config 3 = hidden = value_hidden = 256."""
    rng = np.random.default_rng(seed)
    with open(path, "wb") as f:
        f.write(struct.pack("<Q", activation - 1))
        for i, o in layer_dims(**dims):
            k = 1.0 / np.sqrt(i)
            f.write(struct.pack("<II", i, o))
            f.write((rng.random(o) * 2 * k - k).astype("<f4").tobytes())
            f.write((rng.random((o, i)) * 2 * k - k).astype("<f4").tobytes())


def flops_per_leaf(hidden=256, value_hidden=256, nnz_pokemon=12, nnz_active=45):
    """main value path + embeddings recomputed per leaf."""
    main = 2 * (768 * hidden + hidden * hidden + hidden * value_hidden + value_hidden)
    emb = 10 * 2 * (nnz_pokemon * 128 + 128 * 59) + 2 * 2 * (nnz_active * 128 + 128 * 83)
    return main, emb
