# quick manual GPU check (not a pytest file): parity on a small batch + crude timing
import sys, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import oracle_lib as O
from oak_amd.engine import Context
ctx = Context(0)
n = 2048
b, d, p, r = O.make_random_ou_batch(n)
got = ctx.rollout(b, d, r, p, max_steps=1000, return_state=True)
ob, od, op = b.copy(), d.copy(), p.copy()
oout, osteps = O.rollout_batch(ob, od, r, op, max_steps=1000, threads=8)
print('steps eq', (got['steps'] == osteps).all(), 'results eq', (got['results'] == oout).all(),
      'state eq', (got['battles'] == ob).all(), 'dur eq', (got['durations'] == od).all(), 'prng eq', (got['prng'] == op).all())
if not (got['battles'] == ob).all():
    bad = np.nonzero((got['battles'] != ob).any(axis=1))[0]
    print('bad lanes', bad[:10], 'of', len(bad))
    i = bad[0]
    print('lane', i, 'steps gpu/cpu', got['steps'][i], osteps[i], 'diff bytes', np.nonzero(got['battles'][i] != ob[i])[0][:20])
n = 65536
b, d, p, r = O.make_random_ou_batch(n)
for _ in range(3):
    t = time.time(); got = ctx.rollout(b, d, r, p, max_steps=1000); dt = time.time() - t
    print('host-path rollout 65536: %.3fs  steps %d  -> %.1f M steps/s (PCIe+alloc inclusive)' % (dt, got['steps'].sum(), got['steps'].sum() / dt / 1e6))
