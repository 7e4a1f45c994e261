"""Manual GPU tool: error of the three main-net kernels (fp16 pairs / bf16 triples / fp32 MFMA) against a float64 evaluation, on the
edited networks of tests/test_gpu_leafnet.py::test_fp16_pair_main_net_scales (checker: the numpy oracle, like the tests)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
sys.path.insert(0, "oracle")
import test_gpu_leafnet as T  # noqa: E402
import nn_oracle as NN  # noqa: E402
from oak_amd.engine import Context, Network  # noqa: E402

ctx = Context(0)
cases = sys.argv[1:] or ["plain", "late12", "late12c", "late60c"]
for case in cases:
    dst = os.path.join(tempfile.mkdtemp(), "n.battle.net")
    base = NN.Net(T.NET256)
    aod, pod, sd = base.aod, base.pod, base.side_dim
    cols = np.array([s * sd + (1 + aod) + q * (1 + pod) + 1 + o for s in range(2) for q in range(5) for o in range(pod)])
    sh = 12 if "12" in case else 60 if "60" in case else 0

    def edit(i, b, W):
        if i == 1 and sh:
            return b * np.float32(2.0 ** sh), W * np.float32(2.0 ** sh)
        if i == 4 and case.endswith("c"):
            W = W.copy()
            W[:, cols] *= np.float32(2.0 ** -sh)
        return b, W
    T._rewrite_net(T.NET256, dst, edit)
    net, onet = Network(ctx, path=dst), NN.Net(dst)
    b, d = T._midgame_states(300, 30, 999)
    out = {}
    emb = None
    for mode in ("pair", "split", "fp32"):
        net.set_main_precision(mode)
        v, e = net.value_inference(b, d, return_embedding=True)
        emb = e if emb is None else emb
        out[mode] = v
    with np.errstate(over="ignore"):
        ref = np.array([T._main_value_f64(onet, emb[i]) for i in range(b.shape[0])])
    print(case, "default", net.main_precision(), {m: float(np.abs(out[m] - ref).max()) for m in out}, "rowmax ratio", float(np.median(np.abs(emb[:, 64:]).max(axis=1) / np.abs(emb[:, :64]).max(axis=1))))
    net.close()
