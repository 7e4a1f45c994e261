#!/bin/bash
# VALU issue budget of one rollout launch (serialised by the profiler): is the kernel issue-bound?
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/pmcv; mkdir -p gpurun_out/pmcv
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmcv/a -- python3 tools/launch_caps.py 1000 > gpurun_out/pmcv/a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/pmcv/a/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    if 'rollout' in r['Kernel_Name']:
        acc[r['Kernel_Name'][:60] + ' grid=' + r['Grid_Size']][r['Counter_Name']] += float(r['Counter_Value'])
tot = collections.defaultdict(float)
for k, v in acc.items():
    print(k, dict(v))
    for c, x in v.items(): tot[c] += x
print('TOTAL', dict(tot))
PY
tail -1 gpurun_out/pmcv/a.log
