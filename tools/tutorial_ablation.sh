#!/bin/bash
# VERDICT r4 #1c: the TUTORIAL's frames-per-game statistic (80.17) under each of the engine's four unverifiable choices (variant name =
# OAK_MULTIHIT_ROLL_FIRST, OAK_PSYWAVE_SHOWDOWN, OAK_COUNTER_SHOWDOWN, OAK_ACCURACY_LAST; the product is 1100), same team pairs and
# seeds for every variant.   here: tools/engine_variants.sh build      GPU box: tools/tutorial_ablation.sh [games] [variants]
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
N=${1:-768}; V=${2:-"1100 1110 0101 0000"}
rm -f gpurun_out/ablation.jsonl
for v in $V; do
  if [ "$v" = 1100 ]; then unset OAKGPU_LIB; else export OAKGPU_LIB=$PWD/prof_build/liboakgpu_v$v.so; fi
  echo "== variant $v" >> gpurun_out/ablation.err
  VARIANT=$v timeout -k 10 ${PER_VARIANT_TIMEOUT:-600} python3 tools/tutorial_stats.py length $N 64 2>> gpurun_out/ablation.err | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); d['variant'] = '$v'; print(json.dumps(d))" >> gpurun_out/ablation.jsonl
done
python3 - <<'PY'
import json
rows = [json.loads(l) for l in open('gpurun_out/ablation.jsonl')]
base = rows[0]
for r in rows:
    same = sum(int(a == b) for a, b in zip(r['lengths'], base['lengths']))
    print(r['variant'], 'games', r['games'], 'mean %.2f +- %.2f' % (r['mean'], r['se']), 'median', r['median'], '>=300:', r['games_of_300_frames_or_more'],
          'W/L/T', r['results_win_lose_tie'], 'games with the product\'s length: %d' % same, '%.0f s' % r['seconds'])
PY
