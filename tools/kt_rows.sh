#!/bin/bash
# kernel trace of one 1080p x 256 spp render with / without the sampler rows (flags 0 / 1024)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for f in 0 1024; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kt_rows_$f -o run -- python3 $R/tools/pmc_run.py $f 1920 1080 256 > $R/gpurun_out/kt_rows_$f.log 2>&1
  echo "flags $f"; grep -h "k_sampler_tables\|k_wf_shade\|k_wf_post\|k_wf_generate" $R/gpurun_out/kt_rows_$f/*kernel_stats.csv $R/gpurun_out/kt_rows_$f/*/*kernel_stats.csv 2>/dev/null | cut -c1-160
done
