#!/bin/bash
# The engine's four compile-time switches (OAK_MULTIHIT_ROLL_FIRST, OAK_PSYWAVE_SHOWDOWN, OAK_COUNTER_SHOWDOWN, OAK_ACCURACY_LAST: DESIGN 0;
# a variant is named by their values in that order, the default build is 1100) in their OTHER settings:
# builds a variant of the product library and of the CPU checker with the same flags and holds them to each other with the
# move-coverage and parity tests -- every variant must be bit-exact GPU == checker, like the default.
#   here (no GPU):   tools/engine_variants.sh build       -> prof_build/liboakgpu_v<M><P>.so, prof_build/liboracle_v<M><P>.so
#   on the GPU box:  gpurun -- tools/engine_variants.sh   -> runs the tests against each variant
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p prof_build
VARIANTS="${VARIANTS:-0000 0100 1000 1110 0101 0111}"
if [ "$1" = build ]; then
  for v in $VARIANTS; do
    F="-DOAK_MULTIHIT_ROLL_FIRST=${v:0:1} -DOAK_PSYWAVE_SHOWDOWN=${v:1:1} -DOAK_COUNTER_SHOWDOWN=${v:2:1} -DOAK_ACCURACY_LAST=${v:3:1}"
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $F -c oak_amd/csrc/oakgpu.hip -o prof_build/oakgpu_v$v.o 2>/dev/null   # (the only unit that holds the engines)
    hipcc --offload-arch=gfx950 -fPIC -shared -o prof_build/liboakgpu_v$v.so prof_build/oakgpu_v$v.o build/obj/collective.o build/obj/leafnet.o build/obj/pkmn_shim.o build/obj/search_host.o build/obj/selfplay.o
    gcc -O3 -march=x86-64-v3 -fPIC $F -shared -o prof_build/liboracle_v$v.so oracle/gen1_engine.c oracle/oak_host.c oracle/nn_host.c -lpthread -lm
    echo "built variant $v"
  done
  exit 0
fi
for v in $VARIANTS; do
  echo "== variant MULTIHIT_ROLL_FIRST=${v:0:1} PSYWAVE_SHOWDOWN=${v:1:1} COUNTER_SHOWDOWN=${v:2:1} ACCURACY_LAST=${v:3:1}"
  OAKGPU_LIB=$PWD/prof_build/liboakgpu_v$v.so ORACLE_SO=$PWD/prof_build/liboracle_v$v.so timeout -k 10 500 \
    python3 -m pytest tests/test_gpu_move_coverage.py tests/test_gpu_parity.py -x -q -m gpu -k "not bench and not rehearsal" 2>&1 | tail -2
done
