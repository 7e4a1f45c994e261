#!/bin/bash
# Region profile of the rollout kernel.  Separate -DOAKGPU_SITE_PROFILE build, never the product library.
#   here (no GPU):   tools/site_profile.sh build      -> prof_build/liboakgpu_prof.so (git-ignored, travels with gpurun)
#   on the GPU box:  gpurun -- tools/site_profile.sh  -> gpurun_out/site_profile.{json,txt}
set -e
cd "${GRAFT_REPO_ROOT:-$(dirname "$0")/..}"
if [ "$1" = build ] || [ ! -f prof_build/liboakgpu_prof.so ]; then
  mkdir -p prof_build
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DOAKGPU_SITE_PROFILE -o prof_build/liboakgpu_prof.so oak_amd/csrc/oakgpu.hip oak_amd/csrc/leafnet.hip oak_amd/csrc/pkmn_shim.hip oak_amd/csrc/search_host.hip oak_amd/csrc/selfplay.hip oak_amd/csrc/collective.hip
  [ "$1" = build ] && exit 0
fi
mkdir -p gpurun_out
python3 tools/site_profile.py prof_build/liboakgpu_prof.so > gpurun_out/site_profile.json 2> gpurun_out/site_profile.txt
cat gpurun_out/site_profile.txt
