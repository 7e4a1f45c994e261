#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmcleaf
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcleaf/a -- python3 bench.py --workload leaf --steps 3 --warmup 1 > gpurun_out/pmcleaf/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d gpurun_out/pmcleaf/b -- python3 bench.py --workload leaf --steps 3 --warmup 1 > gpurun_out/pmcleaf/b.log 2>&1 || echo b-failed
