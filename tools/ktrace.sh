#!/bin/bash
# usage: tools/ktrace.sh TAG FLAGS [W H SPP] -> gpurun_out/kt_TAG (rocprofv3 kernel trace + stats of one reduced render)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_$1 -o run -- python3 $R/tools/pmc_run.py $2 $3 $4 $5 > $R/gpurun_out/kt_$1.log 2>&1
cat $R/gpurun_out/kt_$1/run_kernel_stats.csv
