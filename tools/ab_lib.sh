#!/bin/bash
# A/B an alternative build of the library: VALU instruction count of one launch + the headline bench.
# usage (on the GPU box): tools/ab_lib.sh prof_build/liboakgpu_X.so
set -e
cd "$GRAFT_REPO_ROOT"
cp oak_amd/liboakgpu.so /tmp/liboakgpu_saved.so
cp "$1" oak_amd/liboakgpu.so
tools/gpu_pmc_valu.sh | grep TOTAL | sed "s/.*SQ_INSTS_SALU.: \([0-9.]*\).*SQ_INSTS_VALU.: \([0-9.]*\).*/SALU \1 VALU \2/"
python3 bench.py --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value']/1e9)"
cp /tmp/liboakgpu_saved.so oak_amd/liboakgpu.so
