#!/bin/bash
# Round-5 measurement run (on the GPU box via gpurun): the driver's bench command, the default bench, rocprofv3 kernel
# stats of the driver command, and separate PMC passes for the rollout group launch and the leaf kernels.
# Outputs under gpurun_out/r05/; tools/summarize_profile_r05.py turns them into profiles/r05_*.
set -e
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/r05
mkdir -p $O; rm -rf $O/stats $O/pmc_*
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
echo "driver bench done"; cut -c1-200 $O/bench_driver.json
python3 bench.py > $O/bench.json 2> $O/bench.err
echo "default bench done"; cut -c1-200 $O/bench.json
# the driver's command (minus the CPU legs) under the profiler: per-kernel durations of rollout AND leaf kernels
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/stats.log 2>&1
echo "kernel trace done"
# PMC passes (own runs, --kernel-trace only): one group launch of 20 batches = the headline step structure
R="python3 bench.py --workload rollout --steps 20 --warmup 0 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $R > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $R > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_sq -- $R > $O/pmc_sq.log 2>&1 || echo "sq pmc pass failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq2 -- $R > $O/pmc_sq2.log 2>&1 || echo "sq2 pmc pass failed"
echo "rollout pmc done"
L="python3 bench.py --workload leaf --steps 10 --warmup 2 --no-cpu-baseline"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_leaf -- $L > $O/pmc_leaf.log 2>&1 || echo "leaf pmc pass failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_leaf_fetch -- $L > $O/pmc_leaf_fetch.log 2>&1 || echo "leaf fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_leaf_write -- $L > $O/pmc_leaf_write.log 2>&1 || echo "leaf write pass failed"
echo "leaf pmc done"
C="python3 bench.py --workload config3 --steps 20 --warmup 5 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c3_fetch -- $C > $O/pmc_c3_fetch.log 2>&1 || echo "config3 fetch pass failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c3_write -- $C > $O/pmc_c3_write.log 2>&1 || echo "config3 write pass failed"
echo "config3 pmc done"
C4="python3 bench.py --workload config4 --steps 4 --warmup 1 --no-cpu-baseline"
BENCH_NO_RANK_SHARE=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_c4_fetch -- $C4 > $O/pmc_c4_fetch.log 2>&1 || echo "config4 fetch pass failed"
BENCH_NO_RANK_SHARE=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_c4_write -- $C4 > $O/pmc_c4_write.log 2>&1 || echo "config4 write pass failed"
echo "config4 pmc done"
timeout -k 10 300 python3 tools/root_steps_sweep.py 32,64,128,256 16,32,64,128,0 10 > $O/root_steps_sweep.jsonl 2> $O/root_steps_sweep.err || echo "root steps sweep failed"
echo "root steps sweep done"
tools/gpu_pmc_rootstep.sh > $O/pmc_rootstep.txt 2>&1 || echo "root step pmc failed"
cp gpurun_out/pmcrs/summary.json $O/divergence_bound.json 2>/dev/null || true
find $O -name "*.csv" | wc -l
