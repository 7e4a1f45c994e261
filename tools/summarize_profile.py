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
            out["rocprof_avg_dispatch_ms"] = float(r["AverageNs"]) / 1e6
            out["rocprof_kernel_calls"] = int(r["Calls"])
            out["rocprof_kernel_total_ms"] = float(r["TotalDurationNs"]) / 1e6
            out["rocprof_kernel"] = r["Name"][:60]
# one bench step = the regrouping rounds' dispatches of k_rollout_queue on one stream: per-step kernel time from the trace
f = one("stats/*/*_kernel_trace.csv")
if f:
    per_stream = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_rollout" in r["Kernel_Name"]:
            per_stream[r.get("Stream_Id", r.get("Queue_Id"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Grid_Size_X"])))
    spans, busy = [], []
    for rows_ in per_stream.values():
        rows_.sort()
        gmax = max(g for _, _, g in rows_)
        cur = None
        for st, en, g in rows_:
            if g == gmax:                      # round 0 opens a step
                if cur:
                    spans.append(cur[1] - cur[0]); busy.append(cur[2])
                cur = [st, en, 0]
            if cur:
                cur[1] = en; cur[2] += en - st
        if cur:
            spans.append(cur[1] - cur[0]); busy.append(cur[2])
    if spans:
        out["rocprof_steps"] = len(spans)
        out["rocprof_avg_step_span_ms"] = sum(spans) / len(spans) / 1e6      # first dispatch start .. last dispatch end
        out["rocprof_avg_step_kernel_ms"] = sum(busy) / len(busy) / 1e6      # sum of the step's dispatch durations


def pmc(sub):
    """Counter totals of the rollout kernel PER STEP (= summed over the step's regrouping dispatches)."""
    f = one("%s/*/*_counter_collection.csv" % sub)
    agg = collections.defaultdict(float)
    grids = collections.Counter()
    if f:
        seen = set()
        for r in csv.DictReader(open(f)):
            if "k_rollout" in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                if r["Dispatch_Id"] not in seen:
                    seen.add(r["Dispatch_Id"]); grids[int(r["Grid_Size"])] += 1
    steps = grids[max(grids)] if grids else 1   # round-0 dispatches (largest grid) = steps
    return {k: v / steps for k, v in agg.items()}


fetch, write, sq = pmc("pmc_fetch"), pmc("pmc_write"), pmc("pmc_sq")
out["pmc"] = {"FETCH_SIZE_KB": fetch.get("FETCH_SIZE"), "WRITE_SIZE_KB": write.get("WRITE_SIZE"), **sq}
if fetch.get("FETCH_SIZE") is not None and write.get("WRITE_SIZE") is not None:
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reads exactly half of a
    # wide coalesced stream (calibrated for 16 B/lane loads: the kernel's state loads are mostly dwordx4) -> x2.
    out["k_rollout_hbm_bytes_per_launch"] = int(2 * fetch["FETCH_SIZE"] * 1024 + write["WRITE_SIZE"] * 1024)
    out["traffic_note"] = "2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes), per step (all regrouping dispatches), serial launches of 65536 playouts"
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
               "note": out["traffic_note"], "valu_wave_insts_per_step": sq.get("SQ_INSTS_VALU"),
               "salu_wave_insts_per_step": sq.get("SQ_INSTS_SALU")}, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("bench", "bench_serial")}, indent=1))
