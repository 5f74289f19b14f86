#!/bin/bash
# Round profile of the bench workload: (1) the bench line itself (with in-run PMC traffic, parity gate, CPU baseline),
# (2) rocprofv3 --kernel-trace --stats of the same command (no child processes under the profiler).
# usage (GPU box): tools/profile_round.sh TAG  -> gpurun_out/TAG_bench.json, gpurun_out/TAG_kernel_stats.csv, gpurun_out/TAG_pmc/
R=$GRAFT_REPO_ROOT
TAG=${1:-r2}
python3 $R/bench.py --steps 4 --warmup 1 --keep-pmc gpurun_out/${TAG}_pmc > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_kt -o run -- python3 $R/bench.py --steps 4 --warmup 1 --no-pmc --no-cpu-baseline --no-roofline --no-secondary > $R/gpurun_out/${TAG}_kt.log 2>&1
cp $R/gpurun_out/${TAG}_kt/*/run_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null || cp $R/gpurun_out/${TAG}_kt/run_kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv
head -8 $R/gpurun_out/${TAG}_kernel_stats.csv
tail -c 400 $R/gpurun_out/${TAG}_bench.json
