#!/bin/bash
# usage: tools/pmc_collect.sh TAG FLAGS  -> gpurun_out/pmc_TAG_{a,b}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
B="SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAIT_ANY"
rocprofv3 --pmc $A --kernel-trace -d $R/gpurun_out/pmc_$1_a -o run -- python3 $R/tools/pmc_run.py $2 > $R/gpurun_out/pmc_$1_a.log 2>&1 && \
rocprofv3 --pmc $B --kernel-trace -d $R/gpurun_out/pmc_$1_b -o run -- python3 $R/tools/pmc_run.py $2 > $R/gpurun_out/pmc_$1_b.log 2>&1
