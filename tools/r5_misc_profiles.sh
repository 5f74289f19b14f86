#!/bin/bash
# Round 5 (GPU box): roctx marker trace of one render, the material-sort A/B on the McLaren-class scene (time + SQ counters of the
# shade kernel), and the megakernel against the wavefront pipeline on the bench scene (what a per-path persistent loop costs).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5e; mkdir -p $O
# 1. marker trace (ranges cover their stage's execution: YART_ROCTX_SYNC=1)
YART_ROCTX_SYNC=1 timeout -k 10 200 rocprofv3 --marker-trace --kernel-trace --stats --output-format csv -d $O/marker -o run -- python3 $R/tools/pmc_run.py 0 1920 1080 16 > $O/marker.log 2>&1
ls -R $O/marker | head -30
# 2. material sort on C5's scene: flag 64 = shade the queue entries in queue order
SCENE=mclaren SIZE=3840x2160x32 timeout -k 10 300 python3 $R/tools/variant_ab.py PRODUCT:FLAGS=0 PRODUCT:FLAGS=64 > $O/ab_mclaren_sort.log 2>&1
cat $O/ab_mclaren_sort.log | cut -c1-300
for fl in 0 64; do
  SCENE=mclaren timeout -k 10 200 bash $R/tools/pmc_collect.sh r5s_$fl $fl
  python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_r5s_${fl}_a $R/gpurun_out/pmc_r5s_${fl}_b > $O/pmc_sq_mclaren_sort_$fl.txt 2>&1
done
rm -rf $R/gpurun_out/pmc_r5s_*
# 3. megakernel vs wavefront on the bench scene
SIZE=1920x1080x8 timeout -k 10 300 python3 $R/tools/variant_ab.py PRODUCT:FLAGS=0 PRODUCT:FLAGS=1 > $O/ab_mega.log 2>&1
cat $O/ab_mega.log | cut -c1-300
