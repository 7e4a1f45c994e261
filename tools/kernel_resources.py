#!/usr/bin/env python3
"""Kernel descriptors of the shipped library (oak_amd/liboakgpu.so): VGPRs, SGPRs, spills, scratch and LDS of every gfx950
kernel, read from the code objects' metadata notes -> profiles/<tag>_kernel_resources.json.
usage: tools/kernel_resources.py [tag]"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
lib = os.path.join(ROOT, "oak_amd", "liboakgpu.so")
out = {}
with tempfile.TemporaryDirectory() as td:
    subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--list", "--type=o", "--input=" + lib], stdout=subprocess.DEVNULL) if False else None
    # the fat binary lives in .hip_fatbin; roc-obj-ls / roc-obj-extract are scripts around the same bundler
    fat = os.path.join(td, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
    data = open(fat, "rb").read()
    # each translation unit contributes one __CLANG_OFFLOAD_BUNDLE__ blob; split and unbundle each
    blobs = [m.start() for m in re.finditer(rb"__CLANG_OFFLOAD_BUNDLE__", data)]
    for k, start in enumerate(blobs):
        end = blobs[k + 1] if k + 1 < len(blobs) else len(data)
        bpath = os.path.join(td, "bundle%d" % k)
        open(bpath, "wb").write(data[start:end])
        co = os.path.join(td, "co%d.elf" % k)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                            "--input=" + bpath, "--output=" + co], capture_output=True, text=True)
        if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
        for m in re.finditer(r"- (?:\.agpr_count:.*?\n\s+)?(.*?)(?=\n\s+- \.|\namdhsa\.target|\Z)", notes, re.S):
            pass
        cur = {}
        for line in notes.splitlines():
            line = line.strip()
            mm = re.match(r"-?\s*\.(\w+):\s*(.*)", line)
            if not mm:
                continue
            key, val = mm.group(1), mm.group(2).strip().strip("'")
            if key in ("agpr_count", "args") and line.startswith("- "):
                if cur.get("name"):
                    out[cur["name"]] = cur
                cur = {}
            if key in ("name", "sgpr_count", "vgpr_count", "agpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size",
                       "group_segment_fixed_size", "max_flat_workgroup_size", "wavefront_size"):
                if key == "name" and "name" in cur and not val.startswith("_Z"):
                    continue       # argument names
                cur[key] = int(val) if val.isdigit() else val
        if cur.get("name"):
            out[cur["name"]] = cur
res = {}
for name, d in out.items():
    if not str(name).startswith("_Z"):
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    res[dem[:140]] = {k: v for k, v in d.items() if k != "name"}
path = os.path.join(ROOT, "profiles", "%s_kernel_resources.json" % tag)
json.dump(res, open(path, "w"), indent=1, sort_keys=True)
for k in sorted(res):
    d = res[k]
    print("%-110s vgpr %3s agpr %3s sgpr %3s  spill v %3s s %3s  scratch %4s B  lds %6s" % (k[:110], d.get("vgpr_count"), d.get("agpr_count"), d.get("sgpr_count"),
          d.get("vgpr_spill_count"), d.get("sgpr_spill_count"), d.get("private_segment_fixed_size"), d.get("group_segment_fixed_size")))
