#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmcleaf
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcleaf/a -- python3 bench.py --workload leaf --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcleaf/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d gpurun_out/pmcleaf/b -- python3 bench.py --workload leaf --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcleaf/b.log 2>&1 || echo b-failed
python3 - <<'PY'
import csv, glob, collections
for sub in ("a", "b"):
    fs = glob.glob("gpurun_out/pmcleaf/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs, key=lambda f: __import__("os").path.getmtime(f)))):
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if "embed" in k or "mainnet" in k:
            print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
