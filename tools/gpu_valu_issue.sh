#!/bin/bash
# VALU issue peak of the chip, measured (VERDICT r3 #1a): the sweep of tools/experiments/valu_issue_bench.hip plus PMC passes
# (`--one` configurations, each its own rocprofv3 run) -> gpurun_out/r04/valu_issue*; tools/summarize_valu_issue.py turns
# them into profiles/r04_valu_issue.json, which bench.py reads its VALU issue peak from.
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
B=prof_build/valu_issue_bench
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 -o $B tools/experiments/valu_issue_bench.hip
timeout -k 10 300 $B > $O/valu_issue_sweep.json 2> $O/valu_issue_sweep.err
echo "sweep done: $(grep -c waves_per_simd $O/valu_issue_sweep.json) rows"
# op ids: 4 v_add_u32, 5 v_mul_lo_u32, 6 readlane+writelane, 9 s_add_u32, 10 mix, 11 v_add+s_add
rm -rf $O/valu_issue_pmc; mkdir -p $O/valu_issue_pmc
for cfg in "4 0 1 64" "4 0 2 64" "4 0 4 64" "4 0 8 64" "4 1 1 64" "4 1 4 64" "10 0 4 64" "10 1 4 64" "6 0 4 64" "5 0 4 64" "11 0 4 64" "4 0 4 16"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/valu_issue_pmc/$tag -- $B --one $cfg > $O/valu_issue_pmc/$tag.log 2>&1 || echo "pmc pass $tag failed"
done
echo "pmc passes done: $(find $O/valu_issue_pmc -name '*counter_collection.csv' | wc -l)"
