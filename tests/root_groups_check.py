"""Child process of tests/test_gpu_parity.py::test_root_groups_pipeline_gives_the_unpipelined_per_root_results (GPU box only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
torch.cuda.init()
import oracle_lib as O  # noqa: E402
from oak_amd import dist as D  # noqa: E402
from oak_amd.engine import Context  # noqa: E402

roots, reps, K = 12, 256, 2
rb, rd, _, rr = O.make_random_ou_batch(roots, seed0=0x0A4B00000000)
B, Dd, R = np.repeat(rb, reps, axis=0), np.repeat(rd, reps, axis=0), np.repeat(rr, reps)
rng = np.random.default_rng(9)
prng0 = rng.integers(0, 256, (roots * reps, 8), dtype=np.uint8)
prng0[:, 0] |= 1
# oracle: K steps over all lanes, the choice-RNG streams continuing from step to step
op = prng0.copy()
exp_means, exp_res, exp_steps, exp_total = [], None, None, 0
for _ in range(K):
    ob, od = B.copy(), Dd.copy()
    exp_res, exp_steps = O.rollout_batch(ob, od, R, op, max_steps=1000, prep=True, threads=8)
    exp_total += int(exp_steps.sum())
    t = exp_res & 15
    v = np.where(t == 1, 1.0, np.where(t == 2, 0.0, 0.5)).astype(np.float32)
    exp_means.append(v.reshape(roots, reps))
dev = torch.device("cuda", 0)
for groups, ordered, ppl in ((1, True, 2), (3, False, 1), (5, False, 2), (3, True, 1)):
    def make_context(ppl=ppl):
        c = Context(0)
        c.set_playouts_per_lane(ppl)        # both kernels: the playout queue (2) and one lane per playout (1, what bench.py's groups use)
        return c
    tb, td, tr, tp = (torch.from_numpy(x.copy()).to(dev) for x in (B, Dd, R, prng0))
    rg = D.RootGroups(make_context, dev, tb, td, tr, tp, roots, reps, groups)
    rg.run(K, ordered=ordered, keep=True)
    torch.cuda.synchronize(dev)
    assert (rg.results.cpu().numpy() == exp_res).all() and (rg.steps_out.cpu().numpy() == exp_steps.astype(np.int32)).all()
    assert (tp.cpu().numpy() == op).all()
    assert int(rg.total.sum().item()) == exp_total          # the turn-steps of both steps, summed on the device
    for k in range(K):
        got = D.assemble_group_means(roots, 1, groups, [G["history"][k] for G in rg.groups])
        # a root's values are 0 / 0.5 / 1: their sum is exact in any order, so the mean is THE mean
        assert (got == exp_means[k].astype(np.float64).mean(axis=1).astype(np.float32)).all(), (groups, ordered, k)
    rg.close()
    del tb, td, tr, tp
print("root groups ok")
