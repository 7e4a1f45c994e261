#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes. Outputs under gpurun_out/<tag>/.
set -e
TAG=${1:-r01}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
cat $O/bench.json
python3 bench.py --streams 1 --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_serial.json 2>> $O/bench.err
cat $O/bench_serial.json
# same command as the headline bench (minus the CPU leg) under the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu-baseline > $O/stats.log 2>&1
# HBM traffic and issue counters of the rollout kernel: separate PMC passes, serial launches of the headline schedule
# (one step = the regrouping rounds' dispatches of k_rollout_queue; the summary sums them per step)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/pmc_sq -- python3 bench.py --streams 1 --steps 4 --warmup 1 --no-cpu-baseline > $O/pmc_sq.log 2>&1 || echo "sq pmc pass failed"
find $O -name "*.csv" | head -40
