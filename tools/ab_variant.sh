#!/bin/bash
# A/B of a variant build of the library (tools/engine_variants.sh) against the product: SQ_INSTS_VALU / SALU of one rollout launch
# (exact counts) + the headline bench at 20 and 160 steps.   usage (GPU box): tools/ab_variant.sh prof_build/liboakgpu_v00.so
cd "$GRAFT_REPO_ROOT"
for lib in product "$@"; do
  if [ "$lib" = product ]; then unset OAKGPU_LIB; else export OAKGPU_LIB=$PWD/$lib; fi
  echo "== $lib"
  tools/gpu_pmc_valu.sh | grep TOTAL | sed "s/.*SQ_INSTS_SALU.: \([0-9.]*\).*SQ_INSTS_VALU.: \([0-9.]*\).*/SALU \1 VALU \2/"
  for s in 20 160 160; do
    python3 bench.py --workload rollout --steps $s --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench steps $s', round(d['value']/1e9,3))"
  done
done
