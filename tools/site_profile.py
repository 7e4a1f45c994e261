"""Region profile of the rollout kernel (manual GPU tool, not part of the product or the tests).

Builds nothing itself: tools/site_profile.sh compiles oak_amd/csrc with -DOAKGPU_SITE_PROFILE into
gpurun_out/liboakgpu_prof.so; this script loads THAT library (never the product one), runs one full-size
rollout launch and prints, per region of gen1_regs.hpp, the share of wave time and the SIMT efficiency
(active lanes per pass / 64)."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, ".")
from oak_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from oak_amd.engine import Context  # noqa: E402

NAMES = ["refill", "step", "legal_draw", "order", "exec_move", "switch_in", "before_move", "exec_selected_pre",
         "run_move", "gates_hit", "status_bodies", "damage", "secondary_apply", "faint_residual", "publish",
         "calc_damage", "apply_hits", "damage_tail", "heavy_switch", "h_conversion", "h_haze", "h_heal", "h_mimic", "h_poison",
         "h_substitute", "h_transform", "h_bide", "h_sleep", "h_disable"]
ctx = Context(0)
lib, h = ctx.lib, ctx.handle
dev = torch.device("cuda", 0)
ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
ctx.ensure_ou_pools()
n = int(os.environ.get("N", 65536))
T = lambda *s, dt=torch.uint8: torch.empty(s, dtype=dt, device=dev)
battles, durations, prng, rin, rout = T(n, 384), T(n, 8), T(n, 8), T(n), T(n)
steps, values = T(n, dt=torch.int32), T(n, dt=torch.float32)
P = lambda t: C.c_void_p(t.data_ptr())
_lib.check(lib.oakgpu_random_ou_battles_dev(h, C.c_uint64(0x0A4B00000000 + int(os.environ.get("SEED_OFF", 0))), n, P(battles), P(durations), P(prng), P(rin)))
torch.cuda.synchronize()
buf = (C.c_ulonglong * 128)()
lib.oakgpu_site_profile.argtypes = [C.c_void_p, C.c_int]
lib.oakgpu_site_profile(buf, 1)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ev0.record()
_lib.check(lib.oakgpu_rollout_dev(h, P(battles), P(durations), P(rin), P(prng), n, 1000, 0, P(rout), P(steps), P(values), None, None))
ev1.record()
torch.cuda.synchronize()
print("launch ms", ev0.elapsed_time(ev1), file=sys.stderr)
lib.oakgpu_site_profile(buf, 0)
total_steps = int(steps.sum().item())
rows = []
step_cycles = buf[1 * 4 + 2]
for i, nm in enumerate(NAMES):
    passes, lanes, cyc, lanecyc = buf[4 * i:4 * i + 4]
    rows.append({"region": nm, "passes": passes, "lanes_per_pass": lanes / max(passes, 1), "cycles_per_pass": cyc / max(passes, 1),
                 "share_of_step_cycles": cyc / max(step_cycles, 1), "simt_eff": lanecyc / max(64 * cyc, 1)})
print(json.dumps({"turn_steps": total_steps, "n": n, "rows": rows}))
for r in rows:
    print("%-18s passes %10d  lanes/pass %5.1f  cyc/pass %8.0f  share %6.3f  eff %5.3f" % (
        r["region"], r["passes"], r["lanes_per_pass"], r["cycles_per_pass"], r["share_of_step_cycles"], r["simt_eff"]), file=sys.stderr)
