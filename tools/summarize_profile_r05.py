#!/usr/bin/env python3
"""Summarise gpurun_out/r05/ (tools/gpu_profile_r05.sh) into profiles/r05_*.{csv,json} and refresh profiles/traffic.json."""
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "r05")
dst = os.path.join(ROOT, "profiles")


def one(pattern):
    fs = glob.glob(os.path.join(src, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def last_json(path):
    if not path or not os.path.exists(path):
        return None
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1]) if lines else None


out = {}
for name in ("bench_driver", "bench"):
    j = last_json(os.path.join(src, name + ".json"))
    if j:
        json.dump(j, open(os.path.join(dst, "r05_%s.json" % name), "w"), indent=1)
        out[name + "_value"] = j.get("value")

f = one("stats/**/*kernel_stats.csv")
if f:
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(dst, "r05_kernel_stats.csv"), "w") as g:
        w = csv.writer(g)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    out["kernel_avg_us"] = {r["Name"][:60]: float(r["AverageNs"]) / 1e3 for r in rows if r["Name"].startswith(("oak::", "void oak::"))}
f = one("stats/**/*kernel_trace.csv")
if f:   # the timed group launch = the largest-grid k_rollout_queue dispatch of the run
    rows = [r for r in csv.DictReader(open(f)) if "k_rollout_queue" in r["Kernel_Name"]]
    durs = sorted(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, int(r["Grid_Size_X"])) for r in rows)
    if durs:
        big = max(g for _, g in durs)
        d = [t for t, g in durs if g == big]
        out["rocprof_group_launch_ms"] = {"calls": len(d), "avg": sum(d) / len(d), "max": max(d), "grid_x": big}


def pmc(sub, match):
    f = one(sub + "/**/*counter_collection.csv")
    if not f:
        return None
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if match in r["Kernel_Name"]:
            per[r["Kernel_Name"][:48] + " grid=" + r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: {"launches": len(v), "mean": sum(v) / len(v), "max": max(v)} for c, v in d.items()} for k, d in per.items()}


roll = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    p = pmc(sub, "k_rollout_queue")
    if p:
        for k, v in p.items():
            roll.setdefault(k, {}).update(v)
out["rollout_pmc"] = roll
leaf = {}
for sub in ("pmc_leaf", "pmc_leaf_fetch", "pmc_leaf_write"):
    p = pmc(sub, "oak::k_")
    if p:
        for k, v in p.items():
            if "rollout" in k or "random_ou" in k:
                continue
            leaf.setdefault(k, {}).update(v)
for k, v in leaf.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "SQ_BUSY_CYCLES" in v:
        # MFMA_BUSY counts cycles per SIMD summed over the chip; BUSY_CYCLES counts per SE/XCD: report the raw pair and the
        # per-kernel duration-based fraction computed in DESIGN.md from the kernel's duration and 1024 SIMDs
        v["mfma_busy_cycles_per_simd"] = v["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / 1024.0
        if "GRBM_GUI_ACTIVE" in v:  # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = the kernel's duration in GPU cycles (under the profiler)
            v["kernel_cycles"] = v["GRBM_GUI_ACTIVE"]["mean"] / 8.0
            v["mfma_busy_frac"] = v["mfma_busy_cycles_per_simd"] / v["kernel_cycles"]
json.dump(leaf, open(os.path.join(dst, "r05_leaf_pmc.json"), "w"), indent=1)


def hbm_bytes(v):   # guide: FETCH_SIZE (KB) reports half of the bytes of wide coalesced reads on gfx950 -> x2; WRITE_SIZE exact
    return (2 * v.get("FETCH_SIZE", {"mean": 0})["mean"] + v.get("WRITE_SIZE", {"mean": 0})["mean"]) * 1024


leaf_tr = {}
lk = {k: v for k, v in leaf.items() if ("k_embed_both<false>" in k or "k_mainnet" in k) and "FETCH_SIZE" in v and "WRITE_SIZE" in v}
if len(lk) >= 2:
    tot = sum(hbm_bytes(v) for v in lk.values())
    leaf_tr = {"leaf_hbm_bytes_per_call_65536": tot, "leaf_hbm_bytes_per_leaf": tot / 65536,
               "leaf_source": "profiles/r05_leaf_pmc.json: k_embed_both<false> + the main-net kernel, 2 x FETCH_SIZE + WRITE_SIZE (KB), one 65536-leaf value_inference call"}
c3 = {}
for sub in ("pmc_c3_fetch", "pmc_c3_write"):
    p = pmc(sub, "oak::k_")
    if p:
        for k, v in p.items():
            if "random_ou" in k or "build_table" in k:
                continue
            c3.setdefault(k, {}).update(v)
c3_tr = {}
c3k = {k: v for k, v in c3.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v and any(t in k for t in ("k_rollout_staged", "k_party_tags", "k_embed_both<true>", "k_mainnet"))}
if c3k:
    tot = sum(hbm_bytes(v) for v in c3k.values())
    c3_tr = {"config3_hbm_bytes_per_step_65536": tot, "config3_hbm_bytes_per_lane_turn": tot / 65536,
             "config3_source": "gpurun_out/r05 pmc_c3_* passes: mean per launch of k_rollout_staged + k_party_tags + k_embed_both<true> + the main-net kernel, 2 x FETCH_SIZE + WRITE_SIZE",
             "config3_kernels": {k: hbm_bytes(v) for k, v in c3k.items()}}
out["config3_pmc"] = c3
# configs[3]: the sliced search steps' launches (k_root_step, one per step) over a profiled 4-step run
c4 = {}
for sub in ("pmc_c4_fetch", "pmc_c4_write"):
    p = pmc(sub, "k_root_step<")
    if p:
        for k, v in p.items():
            c4.setdefault(k, {}).update(v)
c4_tr = {}
c4j = last_json(os.path.join(src, "pmc_c4_fetch.log"))
c4k = {k: v for k, v in c4.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}
if c4k and c4j:
    k, v = max(c4k.items(), key=lambda kv: kv[1]["FETCH_SIZE"]["launches"])
    steps_per_step = c4j["value"] * c4j["ms_per_step"] / 1e3
    c4_tr = {"config4_hbm_bytes_per_step_launch": hbm_bytes(v), "config4_hbm_bytes_per_turn_step": hbm_bytes(v) / steps_per_step,
             "config4_source": "gpurun_out/r05 pmc_c4_* passes: mean per launch of %s (2 x FETCH_SIZE + WRITE_SIZE) / %.0f turn-steps per search step" % (k, steps_per_step)}
out["config4_pmc"] = c4
sw = os.path.join(src, "root_steps_sweep.jsonl")
if os.path.exists(sw):
    rows = [json.loads(l) for l in open(sw) if l.startswith("{")]
    json.dump({"what": "configs[3] in slices (tools/root_steps_sweep.py, oakgpu_root_steps / k_root_step, raw C ABI): ms per search step and turn-steps/s by roots in flight "
                       "on ONE GPU and by slice length (0 = every step runs its playouts to terminal).  32 roots = one rank's share at 8 GPUs.",
               "rows": rows}, open(os.path.join(dst, "r05_root_steps_sweep.json"), "w"), indent=1)

# the group launch (largest grid) is the headline's dominant kernel: per-launch traffic and instruction counts
drv = last_json(os.path.join(src, "bench_driver.json"))
grids = sorted(roll.items(), key=lambda kv: -int(kv[0].split("grid=")[1]))
if grids and drv:
    k, v = grids[0]
    steps = drv["roofline"]["turn_steps_per_launch"]
    tr = {}
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        # guide: FETCH_SIZE (KB) reports half of the bytes of wide coalesced reads on gfx950 -> x2; WRITE_SIZE exact
        hbm = (2 * v["FETCH_SIZE"]["max"] + v["WRITE_SIZE"]["max"]) * 1024
        tr["k_rollout_hbm_bytes_per_launch"] = hbm
        tr["k_rollout_hbm_bytes_per_turn_step"] = hbm / steps
    if "SQ_INSTS_VALU" in v:
        tr["valu_wave_insts_per_turn_step"] = v["SQ_INSTS_VALU"]["max"] / steps
        tr["salu_wave_insts_per_turn_step"] = v.get("SQ_INSTS_SALU", {"max": 0})["max"] / steps
    if "SQ_THREAD_CYCLES_VALU" in v and "SQ_INSTS_VALU" in v:
        # active lanes per VALU wave-instruction (of 64): the divergence tax in one number.  Units uncalibrated by the guide;
        # read as thread-instructions / wave-instructions it agrees with the in-kernel region profile (tools/site_profile.py)
        tr["valu_active_lanes_per_wave_inst"] = v["SQ_THREAD_CYCLES_VALU"]["max"] / v["SQ_INSTS_VALU"]["max"]
    tr["source"] = "profiles/r05_summary.json (gpurun_out/r05 PMC passes: one group launch of 20 x 65536 playouts, %d turn-steps)" % steps
    tr["note"] = "HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KB -> bytes; gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md), per group launch"
    tr.update(leaf_tr)
    tr.update(c3_tr)
    tr.update(c4_tr)
    db = os.path.join(src, "divergence_bound.json")
    if os.path.exists(db):      # tools/gpu_pmc_rootstep.sh: k_root_step on 256 roots x 4,096 playouts in slices of 64 (configs[3]'s input)
        dj = json.load(open(db))
        if "roots256" in dj:
            tr["config4_valu_wave_insts_per_turn_step"] = dj["roots256"]["valu_wave_insts_per_turn_step"]
            tr["config4_valu_active_lanes_per_wave_inst"] = dj["roots256"]["active_lanes_per_valu_inst"] * 4.0
            tr["config4_valu_source"] = "gpurun_out/r05/divergence_bound.json (tools/gpu_pmc_rootstep.sh: SQ_INSTS_VALU of four k_root_step launches, 256 roots x 4,096 playouts, slices of 64)"
    json.dump(tr, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
    out["traffic"] = tr
json.dump(out, open(os.path.join(dst, "r05_summary.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k not in ("rollout_pmc",)}, indent=1)[:3000])
