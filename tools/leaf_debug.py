"""Manual GPU tool: where does the battle embedding differ from the oracle's / between two calls?  usage: tools/leaf_debug.py <lib.so>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from oak_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
import nn_oracle as NN
import oracle_lib as O
from oak_amd.engine import Context, Network

ctx = Context(0)
path = os.path.join(ROOT, "tests", "golden", "net_default.battle.net")
net, onet = Network(ctx, path=path), NN.Net(path)
b, d, p, r = O.make_random_ou_batch(150, seed0=8080)
O.rollout_batch(b, d, r, p, max_steps=12, threads=4)
v1, e1 = net.value_inference(b, d, return_embedding=True)
v2, e2 = net.value_inference(b, d, return_embedding=True)
print("two calls: value diff", np.abs(v1 - v2).max(), "embedding diff", np.abs(e1 - e2).max())
oe = np.stack([NN.battle_embedding(onet, b[i], d[i]) for i in range(b.shape[0])])
for name, e in (("call 1", e1), ("call 2", e2)):
    bad = np.argwhere(np.abs(e - oe) > 2e-5)
    print(name, "entries off by > 2e-5:", len(bad))
    for leaf, col in bad[:40]:
        side, c = divmod(int(col), 384)
        blk = "active" if c < 84 else "slot %d" % (1 + (c - 84) // 60)
        off = c if c < 84 else (c - 84) % 60
        print("  leaf %3d side %d %-7s col %2d: got %.6f want %.6f" % (leaf, side, blk, off, e[leaf, col], oe[leaf, col]))
rr = np.array([O.LIB.oracle_result_from_state(O.ptr(b[i])) for i in range(b.shape[0])], dtype=np.uint8)
c1, n1 = ctx.choices(b, rr, 0)
c2, n2 = ctx.choices(b, rr, 1)
for rep in range(3):
    vp, l1, l2 = net.value_policy_inference(b, d, c1, n1, c2, n2)
    pl = net.value_inference(b, d)
    bad = np.argwhere(np.abs(vp - pl) > 0)
    print("policy call vs plain call: rows that differ:", bad.ravel().tolist(), "max", np.abs(vp - pl).max(), "| plain vs first call", np.abs(pl - v1).max(), "policy vs first", np.abs(vp - v1).max())
# stress: the same call again and again, every output compared with the first call's
nbad = 0
for rep in range(int(os.environ.get("STRESS", "300"))):
    v, e = net.value_inference(b, d, return_embedding=True)
    if not (v == v1).all() or not (e == e1).all():
        nbad += 1
        bad = np.argwhere(e != e1)
        print("rep %d: %d values differ (rows %s), %d embedding entries differ; first: %s" % (rep, int((v != v1).sum()), np.argwhere(v != v1).ravel().tolist()[:8], len(bad),
              [(int(l), int(c) // 384, (int(c) % 384)) for l, c in bad[:6]]))
print("stress: %d bad repetitions" % nbad)
