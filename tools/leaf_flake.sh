#!/bin/bash
# Manual GPU tool: how often do the leaf tests fail with a given library?  usage: tools/leaf_flake.sh <reps> <name|current> ...
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
cp oak_amd/liboakgpu.so /tmp/liboakgpu_saved.so
reps=$1; shift
for v in "$@"; do
  [ "$v" = current ] && cp /tmp/liboakgpu_saved.so oak_amd/liboakgpu.so || cp prof_build/liboakgpu_$v.so oak_amd/liboakgpu.so
  fails=0
  for i in $(seq $reps); do
    timeout -k 10 200 python3 -m pytest tests/test_gpu_leafnet.py -m gpu -q > gpurun_out/r04/flake_$v.log 2>&1 || { fails=$((fails+1)); grep -h "^FAILED\|^E   *Assert" gpurun_out/r04/flake_$v.log | head -3; }
  done
  echo "$v: $fails failed runs of $reps"
done
cp /tmp/liboakgpu_saved.so oak_amd/liboakgpu.so
