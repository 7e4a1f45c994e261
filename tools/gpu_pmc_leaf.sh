#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmcleaf
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcleaf/a -- python3 bench.py --workload leaf --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcleaf/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d gpurun_out/pmcleaf/b -- python3 bench.py --workload leaf --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmcleaf/b.log 2>&1 || echo b-failed
python3 - <<'PY'
import csv, glob, collections, json, os
out = {}
for sub in ("a", "b"):
    fs = glob.glob("gpurun_out/pmcleaf/%s/**/*counter_collection.csv" % sub, recursive=True)
    if not fs:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in acc.items():
        if "embed" in k or "mainnet" in k:
            o = out.setdefault(k, {})
            o.update({c: round(sum(x) / len(x)) for c, x in v.items()})
            o["avg_duration_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
for k, o in out.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in o:
        # MFMA pipe busy cycles summed over the 1024 SIMDs / (kernel duration x 1024 SIMDs x 2.4 GHz)
        o["mfma_busy_frac"] = o["SQ_VALU_MFMA_BUSY_CYCLES"] / (o["avg_duration_us_under_pmc"] * 1e-6 * 2.4e9 * 1024)
json.dump(out, open("gpurun_out/pmcleaf/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
