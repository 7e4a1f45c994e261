#!/usr/bin/env python3
"""Summarise gpurun_out/<tag>/ (tools/gpu_profile.sh) into profiles/<tag>_*.{csv,json,md}."""
import collections, csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    fs = glob.glob(os.path.join(src, pattern))
    return max(fs, key=os.path.getmtime) if fs else None


out = {}
f = one("stats/*/*_kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    for r in rows:
        if "k_rollout" in r["Name"]:
            out["rocprof_avg_kernel_ms"] = float(r["AverageNs"]) / 1e6
            out["rocprof_kernel_calls"] = int(r["Calls"])
            out["rocprof_kernel"] = r["Name"][:60]


def pmc(sub):
    f = one("%s/*/*_counter_collection.csv" % sub)
    agg = collections.defaultdict(list)
    if f:
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
out["pmc"] = {"FETCH_SIZE_KB": fetch.get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get("WRITE_SIZE"), **sq}
if fetch.get("FETCH_SIZE") is not None and write.get("WRITE_SIZE") is not None:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads exactly half of a
    # wide coalesced stream (calibrated for 16 B/lane loads: the kernel's state loads are mostly dwordx4) -> x2.
    out["k_rollout_hbm_bytes_per_launch"] = int(2 * fetch["FETCH_SIZE"] * 1024 + write["WRITE_SIZE"] * 1024)
    out["traffic_note"] = "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), per launch, serial launches of 65536 playouts"
for name in ("bench.json", "bench_serial.json"):
    p = os.path.join(src, name)
    if os.path.exists(p):
        try:
            out[name[:-5]] = json.loads(open(p).read().strip().splitlines()[-1])
        except Exception as e:  # noqa
            out[name[:-5]] = "unparsed: %s" % e
json.dump(out, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1)
if "k_rollout_hbm_bytes_per_launch" in out:
    json.dump({"k_rollout_hbm_bytes_per_launch": out["k_rollout_hbm_bytes_per_launch"], "source": "profiles/%s_summary.json" % tag,
               "note": out["traffic_note"]}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("bench", "bench_serial")}, indent=1))
