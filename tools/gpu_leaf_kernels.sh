#!/bin/bash
# per-kernel times of the leaf-eval workload (rocprofv3 kernel trace)
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
rm -rf gpurun_out/leafprof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/leafprof -- python3 bench.py --workload leaf --no-cpu-baseline > gpurun_out/leaf.log 2>&1
tail -1 gpurun_out/leaf.log | cut -c1-200
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/leafprof/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:60]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in d.items():
    print("%-62s calls %4d  avg %9.1f us" % (k, len(v), sum(v) / len(v) / 1e3))
PY
