"""Child process of tests/test_gpu_parity.py::test_root_steps_host_class_in_a_torch_process (GPU box only): oak_amd.dist.RootSteps --
torch tensors, the exchange hook (a stand-in that copies: one rank), the pinned host record -- against the oracle's credited aggregates."""
import ctypes as C
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

roots, reps, K, slice_ = 9, 128, 4, 64
rb, rd, _, rr = O.make_random_ou_batch(roots, seed0=0x0A4B00000000)
lane = np.zeros((roots * reps, 8), dtype=np.uint8)
for i in range(roots * reps):
    O.LIB.oracle_fast_prng_seed(O.ptr(lane[i]), C.c_uint64(0xC40000000000 + i))
ref_lane = lane.copy()
cnt, s2, ex = O.root_steps_reference(rb, rd, rr, ref_lane, reps, K, slice_, threads=8)
dev = torch.device("cuda", 0)
ctx = Context(0)
tb, td, tr, tp = (torch.from_numpy(x.copy()).to(dev) for x in (rb, rd, rr, lane))
calls = []


def exchange(send, recv):          # what a one-rank all-gather does, on the context's stream; per = 11 entries (padded)
    calls.append(int(send.numel()))
    recv[:send.numel()].copy_(send)


rs = D.RootSteps(ctx, dev, tb, td, tr, tp, roots, reps, slice=slice_, world=1, exchange=exchange, per=11)
k = 0
while True:
    rs.step(fresh=k < K)
    rec = rs.finish()
    assert rec["count"].shape == (11,) and (rec["count"][:roots] == cnt[k]).all() and (rec["count"][roots:] == 0).all(), k
    assert (rec["sum2"][:roots] == s2[k]).all() and rec["turn_steps"] == ex[k], k
    m = D.credited_means(rec["count"][:roots], rec["sum2"][:roots])
    assert ((m >= 0) & (m <= 1)).all()
    k += 1
    if k >= K and rec["carried"] == 0:
        break
assert calls == [11] * k and rs.turn_steps == int(ex.sum())
assert (tp.cpu().numpy() == ref_lane).all()
rs.close()
ctx.close()
print("root steps ok: %d steps (%d of them drains), %d turn-steps" % (k, k - K, int(ex.sum())))
