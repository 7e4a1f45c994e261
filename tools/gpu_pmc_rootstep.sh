#!/bin/bash
# Sliced against unsliced launches of k_root_step on the headline's heterogeneous workload (2^20 team pairs, one playout each,
# step cap 250): VALU / SALU instructions, wave cycles and wait cycles per turn-step -- is the slices' gain fewer instructions
# (homogeneous waves) or fewer stalls?   usage (GPU box): tools/gpu_pmc_rootstep.sh
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/pmcrs; mkdir -p gpurun_out/pmcrs
for sl in 0 64; do
  SWEEP_MAX_STEPS=250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU --kernel-trace --output-format csv -d gpurun_out/pmcrs/s$sl -- python3 tools/root_steps_sweep.py 1048576 $sl 4 1 > gpurun_out/pmcrs/s$sl.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for sl in (0, 64):
    f = glob.glob('gpurun_out/pmcrs/s%d/**/*counter_collection.csv' % sl, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if 'k_root_step<' in r['Kernel_Name']]
    ids = sorted({int(r['Dispatch_Id']) for r in rows})
    keep = set(ids[-4:])                   # the timed launches (the warm-up ones come first)
    tot = collections.defaultdict(float)
    for r in rows:
        if int(r['Dispatch_Id']) in keep:
            tot[r['Counter_Name']] += float(r['Counter_Value'])
    steps = sum(json.loads(l)['turn_steps_per_step'] for l in open('gpurun_out/pmcrs/s%d.log' % sl) if l.startswith('{')) * 4
    out['slice_%d' % sl] = {'turn_steps': steps, **{k: v for k, v in tot.items()}, **{k + '_per_turn_step': v / steps for k, v in tot.items()}}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmcrs/summary.json', 'w'), indent=1)
PY
