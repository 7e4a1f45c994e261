"""Manual GPU tool: leaf-evaluator soak -- value_inference of the HIP path against the plain-C oracle port (oracle/nn_host.c,
itself checked against nn_oracle.py and the torch-mirror goldens) on whole batches of mid-game states, default / tiny(clamp)
/ 768-256-256-256-1 networks, plain and cached entry points.  usage: leaf_soak.py [n] -- prints the largest |difference|."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
sys.path.insert(0, "oracle")
import nn_oracle as NN  # noqa: E402  (checker only)
import oracle_lib as O  # noqa: E402  (checker only)
from oak_amd.engine import Context, Network  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctx = Context(0)
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
td = tempfile.mkdtemp()
wide = os.path.join(td, "c3.battle.net")
NN.write_random_net(wide, hidden=256, value_hidden=256, seed=7)
worst_all = 0.0
for steps, seed0 in ((0, 0x1EAF0000), (12, 0x1EAF1000), (45, 0x1EAF2000), (110, 0x1EAF3000)):
    b, d, p, r = O.make_random_ou_batch(n, seed0=seed0)
    if steps:
        O.rollout_batch(b, d, r, p, max_steps=steps, threads=16)
    for path in (os.path.join(root, "tests", "golden", "net_default.battle.net"), os.path.join(root, "tests", "golden", "net_tiny.battle.net"), wide):
        net, cnet = Network(ctx, path=path), O.CNet(path)
        got = net.value_inference(b, d)
        exp = cnet.value_inference_batch(b, d, threads=16)
        worst = float(np.abs(got - exp).max())
        worst_all = max(worst_all, worst)
        print("steps %3d  %-24s  leaves %d  max |gpu - oracle| = %.3g" % (steps, os.path.basename(path), n, worst), flush=True)
        net.close()
        cnet.close()
print("leaf soak: max abs difference %.3g over %d evaluations (%s 1e-5)" % (worst_all, 12 * n, "<=" if worst_all <= 1e-5 else "ABOVE"))
sys.exit(0 if worst_all <= 1e-5 else 1)
