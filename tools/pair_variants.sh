#!/bin/bash
# Manual tool: experiment builds of k_mainnet_pair (leafnet.hip with -D switches) -> prof_build/liboakgpu_<name>.so; tools/leaf_variants.sh times them.
#   here (no GPU):  tools/pair_variants.sh build
#   on the GPU box: TESTS=0 tools/leaf_variants.sh pg2 prtz ...
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
mkdir -p prof_build
others=$(ls build/obj/*.o | grep -v leafnet.o)
build() { # name flags...
  local name=$1; shift
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c oak_amd/csrc/leafnet.hip -o prof_build/leafnet_$name.o 2>/dev/null
  hipcc --offload-arch=gfx950 -fPIC -shared -o prof_build/liboakgpu_$name.so prof_build/leafnet_$name.o $others
  rm -f prof_build/leafnet_$name.o
  echo built $name
}
if [ "$1" = exp ]; then
  build pe1 -DOAK_MP_EXP=1 &
  build pe2 -DOAK_MP_EXP=2 &
  build pe4 -DOAK_MP_EXP=4 &
  wait
  build pe8 -DOAK_MP_EXP=8 &
  build pe3 -DOAK_MP_EXP=3 &
  build pe15 -DOAK_MP_EXP=15 &
  wait
  exit 0
fi
build pg2 -DOAK_MP_G=2 &
build prtz -DOAK_MP_SPLIT=1 &
build pnosplit -DOAK_MP_SPLIT=2 &
wait
build pvalu2 -DOAK_MP_VALU=2 &
build pvalu5 -DOAK_MP_VALU=5 &
build prtzg2 -DOAK_MP_SPLIT=1 -DOAK_MP_G=2 &
wait
