#!/bin/bash
# Run on the GPU box (via gpurun): bench + rocprofv3 kernel stats + PMC passes. Outputs under gpurun_out/.
set -e
TAG=${1:-r01}
mkdir -p gpurun_out/$TAG
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 3 > gpurun_out/$TAG/bench.json 2> gpurun_out/$TAG/bench.err
cat gpurun_out/$TAG/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/$TAG/pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/pmc_sq.log 2>&1 || echo "sq pmc pass failed"
find gpurun_out/$TAG -name "*.csv" | head -40
