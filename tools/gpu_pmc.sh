#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d gpurun_out/pmc/a -- python3 tools/launch_caps.py 25 300 1000 > gpurun_out/pmc/a.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmc/b -- python3 tools/launch_caps.py 25 300 1000 > gpurun_out/pmc/b.log 2>&1 || echo pass-b-failed
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --kernel-trace --output-format csv -d gpurun_out/pmc/c -- python3 tools/launch_caps.py 25 300 1000 > gpurun_out/pmc/c.log 2>&1 || echo pass-c-failed
tail -2 gpurun_out/pmc/a.log
