"""Manual GPU tool: large randomized GPU-vs-oracle parity soak (bit-exact final states) beyond the test suite's sizes.
usage: parity_soak.py [batches] [playouts per batch] [engine].  Batches alternate between single launches and group
launches of four (oakgpu_rollout_group), with and without root prep; engine 2 (register-resident, default) or 1 (LDS-resident).
Uses the CPU oracle as the checker, like the tests do."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import oracle_lib as O  # noqa: E402  (checker only)
from oak_amd.engine import Context  # noqa: E402

batches = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
ctx = Context(0)
if len(sys.argv) > 3:
    ctx.set_rollout_engine(int(sys.argv[3]))
bad = 0
total_steps = 0
t0 = time.time()
for k in range(batches):
    b, d, p, r = O.make_random_ou_batch(n, seed0=0x50AC000000 + k * n)
    prep = bool(k & 1)
    if k & 2:   # as a group of four ragged batches through one playout queue
        cuts = [0, n // 5, n // 2, n - 1000, n]
        parts = ctx.rollout_group([(b[i:j], d[i:j], r[i:j], p[i:j]) for i, j in zip(cuts, cuts[1:])], max_steps=1000, prep=prep, return_state=True)
        got = {key: np.concatenate([q[key] for q in parts]) for key in ("steps", "results", "battles", "durations", "prng")}
    else:
        got = ctx.rollout(b, d, r, p, max_steps=1000, prep=prep, return_state=True)
    ob, od, op = b.copy(), d.copy(), p.copy()
    oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, prep=prep, threads=16)
    ok = ((got["steps"] == osteps).all() and (got["results"] == oout).all() and (got["battles"] == ob).all()
          and (got["durations"] == od).all() and (got["prng"] == op).all())
    total_steps += int(osteps.sum())
    bad += 0 if ok else 1
    print("batch %d prep=%d playouts=%d steps=%d %s (%.0fs)" % (k, prep, n, int(osteps.sum()), "OK" if ok else "MISMATCH", time.time() - t0), flush=True)
print("soak:", "ALL BIT-EXACT" if bad == 0 else "%d batches mismatched" % bad, "over", total_steps, "turn-steps")
sys.exit(1 if bad else 0)
