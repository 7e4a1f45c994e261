#!/bin/bash
# Batched refills of the two queue kernels (OAK_REFILL_EVERY / OAK_REFILL_LANES, OAK_ROOT_REFILL_*): variant builds and their A/B.
#   here (no GPU):  tools/refill_variants.sh build "4:64 8:64 8:16 16:16"   -> prof_build/liboakgpu_rf<E>_<L>.so
#   GPU box:        tools/refill_variants.sh run "4:64 ..."                  -> headline at 20 / 160 steps + configs[3] sweep per variant
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p prof_build gpurun_out
V="${2:-4:64 8:64 8:16 16:16}"
if [ "$1" = build ]; then
  for v in $V; do
    E=${v%%:*}; L=${v##*:}
    F="-DOAK_REFILL_EVERY=$E -DOAK_REFILL_LANES=$L -DOAK_ROOT_REFILL_EVERY=$E -DOAK_ROOT_REFILL_LANES=$L"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $F -c oak_amd/csrc/oakgpu.hip -o prof_build/oakgpu_rf${E}_$L.o 2>/dev/null &
  done
  wait
  for v in $V; do
    E=${v%%:*}; L=${v##*:}
    hipcc --offload-arch=gfx950 -fPIC -shared -o prof_build/liboakgpu_rf${E}_$L.so prof_build/oakgpu_rf${E}_$L.o build/obj/collective.o build/obj/leafnet.o build/obj/pkmn_shim.o build/obj/search_host.o build/obj/selfplay.o
    echo "built rf${E}_$L"
  done
  exit 0
fi
for v in product $V; do
  if [ "$v" = product ]; then unset OAKGPU_LIB; tag=product; else E=${v%%:*}; L=${v##*:}; export OAKGPU_LIB=$PWD/prof_build/liboakgpu_rf${E}_$L.so; tag=rf${E}_$L; fi
  for s in 20 160; do
    python3 bench.py --workload rollout --steps $s --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tag headline steps $s', round(d['value']/1e9,3))"
  done
  python3 tools/root_steps_sweep.py 32,256 64 8 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$tag root_steps roots', d['roots'], 'ms', round(d['ms_per_step'],3), 'G', round(d['turn_steps_per_s']/1e9,3))"
done
