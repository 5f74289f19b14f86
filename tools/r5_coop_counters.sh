#!/bin/bash
# Round 5: counters of the lane-cooperative walk (flag 256) next to the binary walk (flag 0), C3 and McLaren-class (GPU box).
R=$GRAFT_REPO_ROOT
for sc in sponza mclaren; do
  for fl in 0 256; do
    SCENE=$sc timeout -k 10 160 bash $R/tools/pmc_collect.sh r5_${sc}_$fl $fl || echo "pmc_collect $sc $fl failed"
    python3 $R/tools/pmc_report.py $R/gpurun_out/pmc_r5_${sc}_${fl}_a $R/gpurun_out/pmc_r5_${sc}_${fl}_b > $R/gpurun_out/r5_pmc_sq_${sc}_$fl.txt 2>&1
    echo "sq $sc $fl done"
  done
done
for fl in 0 256; do
  timeout -k 10 500 bash $R/tools/pmc_mem_path.sh r5_mp_$fl $fl > $R/gpurun_out/r5_mp_$fl.log 2>&1 || echo "mem path $fl failed"
  echo "mem path $fl done"
done
