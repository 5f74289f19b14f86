#!/bin/bash
# Address-path counters of the large kernels over one un-timed step of the bench workload: texture-addresser busy cycles, L1 (TCP)
# accesses / stalls, address-translation (UTCL1) misses. usage (GPU box): tools/pmc_mem_path.sh TAG [FLAGS] -> gpurun_out/TAG_mem_path.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r3}
# (at most two counters of a block per pass: the TA / TCP blocks have few slots; every pass under its own timeout, progress printed)
PASSES=("TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"
        "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "GRBM_GUI_ACTIVE TD_TD_BUSY_sum")
i=0
for ctrs in "${PASSES[@]}"; do
  i=$((i+1))
  echo "pass $i: $ctrs"
  timeout -k 10 150 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_mp_$i -o run -- python3 $R/tools/pmc_run.py ${2:-0} 1920 1080 16 > $R/gpurun_out/${TAG}_mp_$i.log 2>&1 || { echo "pass $i failed"; grep -m2 -i "error\|exceeds" $R/gpurun_out/${TAG}_mp_$i.log | cut -c1-200; }
done
python3 - <<PY > $R/gpurun_out/${TAG}_mem_path.txt
import csv, glob, collections, sys
sys.path.insert(0, "$R")
from bench import kernel_base
agg = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float)
for i in range(1, 8):
    for f in glob.glob("$R/gpurun_out/${TAG}_mp_%d/**/*counter_collection.csv" % i, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = kernel_base(r["Kernel_Name"])
            if k is None: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if i == 1 and r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
for k in sorted(agg, key=lambda k: -dur[k])[:8]:
    print(k, "ms (profiled) %.1f" % dur[k])
    for c, v in sorted(agg[k].items()):
        print("   %-42s %.4e" % (c, v))
PY
cat $R/gpurun_out/${TAG}_mem_path.txt
