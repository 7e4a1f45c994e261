#!/bin/bash
# Manual GPU tool: the driver command's rollout part under a few settings of the queue kernel's migration (environment switches read at
# oakgpu_create).  Round 5, final kernel: every setting within the run-to-run noise (8.5-8.9 G) except donating at 150 turn-steps (7.8 G).
cd $GRAFT_REPO_ROOT
run() { # label envs...
  label=$1; shift
  for rep in 1 2; do
    v=$(env "$@" python3 bench.py --workload rollout --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f' % (d['value']/1e9))")
    echo "$label $v"
  done
}
run base X=1
run adopters64 OAKGPU_MIGRATE_ADOPTERS=64
run adopters96 OAKGPU_MIGRATE_ADOPTERS=96
run adopters192 OAKGPU_MIGRATE_ADOPTERS=192
run window16 OAKGPU_MIGRATE_WINDOW=16
run window40 OAKGPU_MIGRATE_WINDOW=40
run long150 OAKGPU_MIGRATE_STEPS=150
run long300 OAKGPU_MIGRATE_STEPS=300
run ppl3 OAKGPU_PLAYOUTS_PER_LANE=3
run base X=1
