#!/bin/bash
# k_root_step under the counters: how much of the rollout's instruction count is DIVERGENCE between playouts of different team pairs
# and ages?  Same kernel, same number of playouts per launch (2^20), four inputs:
#   hetero_noslice  2^20 different team pairs, one playout each, step cap 250, no slices   (the headline's workload, unsliced)
#   hetero          the same in slices of 64 turn-steps
#   roots256        256 team pairs x 4,096 playouts (BASELINE configs[3]), slices of 64
#   root1           ONE team pair x 2^20 playouts, slices of 64: every lane of every wave plays the same two teams -- what a
#                   perfectly team-binned engine could reach
# VALU / SALU wave-instructions, active lanes per VALU instruction, wave and wait cycles per turn-step -> gpurun_out/pmcrs/summary.json
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/pmcrs; mkdir -p gpurun_out/pmcrs
run() { # tag roots slice reps cap
  SWEEP_MAX_STEPS=$5 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmcrs/$1 -- python3 tools/root_steps_sweep.py $2 $3 4 $4 > gpurun_out/pmcrs/$1.log 2>&1
  SWEEP_MAX_STEPS=$5 python3 tools/root_steps_sweep.py $2 $3 8 $4 > gpurun_out/pmcrs/$1.rate 2>/dev/null
}
run hetero_noslice 1048576 0 1 250
run hetero 1048576 64 1 250
run roots256 256 64 4096 1000
run root1 1 64 1048576 1000
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("hetero_noslice", "hetero", "roots256", "root1"):
    f = glob.glob('gpurun_out/pmcrs/%s/**/*counter_collection.csv' % tag, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'k_root_step<' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    keep = set(ids[-4:])                   # the timed launches (the warm-up ones come first)
    tot = collections.defaultdict(float)
    for r in rows:
        if int(r['Dispatch_Id']) in keep:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
    steps = sum(json.loads(l)['turn_steps_per_step'] for l in open('gpurun_out/pmcrs/%s.log' % tag) if l.startswith('{')) * 4
    rate = [json.loads(l) for l in open('gpurun_out/pmcrs/%s.rate' % tag) if l.startswith('{')][0]
    out[tag] = {"turn_steps_per_s_unprofiled": rate["turn_steps_per_s"], "ms_per_launch_unprofiled": rate["ms_per_step"],
                "valu_wave_insts_per_turn_step": tot["SQ_INSTS_VALU"] / steps, "salu_insts_per_turn_step": tot["SQ_INSTS_SALU"] / steps,
                "active_lanes_per_valu_inst": tot["SQ_THREAD_CYCLES_VALU"] / tot["SQ_INSTS_VALU"] / 4.0 if tot["SQ_INSTS_VALU"] else None,
                "wave_cycles_per_turn_step": tot["SQ_WAVE_CYCLES"] / steps, "wait_any_per_turn_step": tot["SQ_WAIT_ANY"] / steps,
                "turn_steps_profiled": steps}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmcrs/summary.json', 'w'), indent=1)
PY
