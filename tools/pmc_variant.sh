#!/bin/bash
# usage: tools/pmc_variant.sh VARIANT  -> FETCH_SIZE / WRITE_SIZE per kernel of one reduced render (tools/pmc_run.py, full-size textures) with yart_amd/_variants/VARIANT.so
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export YART_LIB=$1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmcv_$1_$c -o run -- python3 $R/tools/pmc_run.py 0 1920 1080 32 > $R/gpurun_out/pmcv_$1_$c.log 2>&1 || exit 1
done
python3 - <<PY
import csv, collections
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(float)
    for r in csv.DictReader(open("$R/gpurun_out/pmcv_$1_%s/run_counter_collection.csv" % c)):
        if r["Counter_Name"] == c and "k_wf_shade" in r["Kernel_Name"]:
            agg["shade"] += float(r["Counter_Value"])
    print("$1", c, {k: round(v * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e9, 2) for k, v in agg.items()}, "GB per render")
PY
