"""Manual GPU tool: value_inference and value_policy_inference of RANDOM network shapes (hidden / value-hidden / policy-hidden
widths 8..256, relu and clamp, ragged batch sizes) against the numpy oracle.  usage: python tools/leaf_fuzz.py [configs=40] [seed=1]"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
sys.path.insert(0, "oracle")
import nn_oracle as NN  # noqa: E402
import oracle_lib as O  # noqa: E402
from oak_amd.engine import Context, Network  # noqa: E402

configs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = Context(0)
td = tempfile.mkdtemp()
worst_v = worst_l = 0.0
for k in range(configs):
    hid, vh, ph = (int(rng.integers(1, 33)) * 8 for _ in range(3))
    act = int(rng.integers(1, 3))
    n = int(rng.integers(1, 300))
    steps = int(rng.choice([0, 5, 20, 60]))
    path = os.path.join(td, "f%d.battle.net" % k)
    NN.write_random_net(path, hidden=hid, value_hidden=vh, policy_hidden=ph, seed=100 + k, activation=act)
    net, onet = Network(ctx, path=path), NN.Net(path)
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x0A4B00000000 + 1000 * k)
    if steps:
        O.rollout_batch(b, d, r, p, max_steps=steps, threads=4)
    res = np.array([O.LIB.oracle_result_from_state(O.ptr(b[i])) for i in range(n)], dtype=np.uint8)
    c1, n1 = ctx.choices(b, res, 0)
    c2, n2 = ctx.choices(b, res, 1)
    for mode in ("pair", "split", "fp32"):
        net.set_main_precision(mode)
        plain = net.value_inference(b, d)
        vals, l1, l2 = net.value_policy_inference(b, d, c1, n1, c2, n2)
        assert np.array_equal(plain, vals), (k, mode)
        ev = el = 0.0
        for i in range(0, n, max(1, n // 24)):
            ov, o1, o2 = NN.value_policy_inference(onet, b[i], d[i], c1[i, :n1[i]], c2[i, :n2[i]])
            ev = max(ev, abs(float(vals[i]) - float(ov)))
            if n1[i]:
                el = max(el, float(np.abs(l1[i, :n1[i]] - o1).max()))
            if n2[i]:
                el = max(el, float(np.abs(l2[i, :n2[i]] - o2).max()))
            assert (l1[i, n1[i]:] == 0).all() and (l2[i, n2[i]:] == 0).all()
        assert ev <= 1e-5 and el <= 2e-5, (k, mode, hid, vh, ph, act, n, ev, el)
        worst_v, worst_l = max(worst_v, ev), max(worst_l, el)
    print("config %2d  hidden %3d value_hidden %3d policy_hidden %3d act %d  n %3d steps %2d  ok" % (k, hid, vh, ph, act, n, steps), flush=True)
    net.close()
print("leaf fuzz: %d random shapes x 3 main-net modes, worst |value - oracle| %.2e, worst |logit - oracle| %.2e" % (configs, worst_v, worst_l))
