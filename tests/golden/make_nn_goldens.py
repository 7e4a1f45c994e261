#!/usr/bin/env python3
"""Generate the MLP golden fixtures from the reference's own torch mirror.

Runs ONLY in the authoring container (needs /root/reference/src/oak/torch.py, which the
reference itself declares equal to its Eigen path: src/oak/lab.py:21-79).  `import oak`
fails there with an ordinary ModuleNotFoundError (the pybind module is unbuilt), so the
file is loaded with importlib after registering a stub module `oak` that carries only the
integer constants torch.py reads (values from cpp/src/pyoak.cc:586-600 /
cpp/include/nn/default-hyperparameters.h).  Outputs (data only):
  tests/golden/net_default.battle.net   seeded default-dim network file (reference writer)
  tests/golden/net_tiny.battle.net      small-dim network (generic-dimension coverage)
  tests/golden/net_256.battle.net       hidden = value_hidden = 256: BASELINE configs[2]'s "3x256 MLP" 
  tests/golden/nn_goldens.npz           inputs + reference outputs of pokemon_net, active_net,
                                        main_net.forward_value_only and main_net.forward (policy logits)
"""
import importlib.util
import io
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("OAK_REFERENCE", "/root/reference")


def load_reference_torch():
    stub = types.ModuleType("oak")
    stub.pokemon_in_dim = 198
    stub.active_in_dim = 427
    stub.pokemon_hidden_dim = 128
    stub.pokemon_out_dim = 59
    stub.active_hidden_dim = 128
    stub.active_out_dim = 83
    stub.hidden_dim = 64
    stub.value_hidden_dim = 32
    stub.policy_hidden_dim = 64
    stub.policy_out_dim = 315
    stub.build_policy_hidden_dim = 128
    stub.build_value_hidden_dim = 128
    stub.species_move_list = []
    for name in ("EncodedBattleFrames", "BuildTrajectories", "OutputBuffer"):
        setattr(stub, name, object)
    sys.modules["oak"] = stub
    spec = importlib.util.spec_from_file_location("oak_ref_torch", os.path.join(REF, "src", "oak", "torch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    T = load_reference_torch()
    out = {}
    rng = np.random.default_rng(20260101)
    for tag, kw, act in (("default", {}, T.Activation.relu),
                         ("tiny", dict(phd=16, ahd=24, pod=8, aod=12, hd=32, vhd=16, pohd=8), T.Activation.clamp),
                         ("256", dict(hd=256, vhd=256), T.Activation.relu)):
        torch.manual_seed({"default": 1234, "tiny": 99, "256": 256}[tag])
        net = T.BattleNetwork(activation=act, **kw)
        with torch.no_grad():   # widen the default init so relu/clamp both saturate and pass
            for p in net.parameters():
                p.mul_(2.0)
        buf = io.BytesIO()
        net.write_parameters(buf)
        with open(os.path.join(HERE, "net_%s.battle.net" % tag), "wb") as f:
            f.write(buf.getvalue())
        k = 16
        # sparse-like inputs in the encoders' value range, plus a few dense rows
        xp = (rng.random((k, 198)) < 0.08).astype(np.float32) * rng.random((k, 198)).astype(np.float32)
        xa = (rng.random((k, 427)) < 0.10).astype(np.float32) * rng.random((k, 427)).astype(np.float32)
        xm = rng.random((k, 2 * net.side_out_dim)).astype(np.float32) * (rng.random((k, 2 * net.side_out_dim)) < 0.7)
        xm = xm.astype(np.float32)
        with torch.no_grad():
            yp = net.pokemon_net.forward(torch.from_numpy(xp)).numpy()
            ya = net.active_net.forward(torch.from_numpy(xa)).numpy()
            ym = net.main_net.forward_value_only(torch.from_numpy(xm)).numpy()
            _, l1, l2 = net.main_net.forward(torch.from_numpy(xm))   # full 315-wide policy logits per side
        out.update({tag + "_xp": xp, tag + "_yp": yp, tag + "_xa": xa, tag + "_ya": ya, tag + "_xm": xm, tag + "_ym": ym,
                    tag + "_l1": l1.numpy(), tag + "_l2": l2.numpy()})
        print(tag, "file bytes", len(buf.getvalue()), "value range", ym.min(), ym.max())
    np.savez_compressed(os.path.join(HERE, "nn_goldens.npz"), **out)


if __name__ == "__main__":
    main()
