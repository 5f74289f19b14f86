#!/bin/bash
# HBM traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate --pmc passes over one
# un-timed step of the bench workload (MI355X_MICROARCH.md, HBM section). Output: gpurun_out/pmc_hbm_{rd,wr}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline $*"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_hbm_rd -o run -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_hbm_rd.log 2>&1 && \
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_hbm_wr -o run -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_hbm_wr.log 2>&1
tail -2 $R/gpurun_out/pmc_hbm_wr.log | cut -c1-300
