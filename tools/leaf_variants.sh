#!/bin/bash
# Manual GPU tool: main-net kernel time of experiment builds (prof_build/liboakgpu_<name>.so; results are NOT checked).
cd "$GRAFT_REPO_ROOT"
cp oak_amd/liboakgpu.so /tmp/liboakgpu_saved.so
for v in "$@"; do
  cp prof_build/liboakgpu_$v.so oak_amd/liboakgpu.so
  timeout -k 10 120 python3 bench.py --workload leaf --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['roofline']['kernel_us'])"
done
cp /tmp/liboakgpu_saved.so oak_amd/liboakgpu.so
