#!/bin/bash
# Manual GPU tool: value_policy_inference timing of experiment builds (prof_build/liboakgpu_<name>.so), each first held to the policy tests.
# usage: tools/policy_variants.sh <name> ...      (TESTS=0 skips the tests)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r04
cp oak_amd/liboakgpu.so /tmp/liboakgpu_saved.so
for v in "$@"; do
  cp prof_build/liboakgpu_$v.so oak_amd/liboakgpu.so
  if [ "${TESTS:-1}" != "0" ]; then
    timeout -k 10 400 python3 -m pytest tests/test_gpu_leafnet.py -m gpu -x -q -k "policy" > gpurun_out/r04/policy_variant_$v.log 2>&1; echo "$v tests: $(tail -1 gpurun_out/r04/policy_variant_$v.log)"
  fi
  for rep in 1 2; do
    timeout -k 10 120 python3 bench.py --workload leaf --steps 20 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['policy']; print('$v', '%.1f M value+policy leaf-evals/s, heads %.1f us' % (p['leaf_evals_per_s']/1e6, p['policy_heads_ms']*1e3), '| value only %.1f M' % (d['value']/1e6))"
  done
done
cp /tmp/liboakgpu_saved.so oak_amd/liboakgpu.so
