#!/bin/bash
# usage (GPU box): tools/pmc_icache.sh TAG -> gpurun_out/pmc_TAG_ic: instruction-cache counters per kernel (one render of tools/pmc_run.py's workload)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$1_ic -o run -- python3 $R/tools/pmc_run.py 0 > $R/gpurun_out/pmc_$1_ic.log 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/pmc_$1_ic/**/run_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:8]:
    req, hit, miss = c.get("SQC_ICACHE_REQ", 0), c.get("SQC_ICACHE_HITS", 0), c.get("SQC_ICACHE_MISSES", 0)
    print(f"{k[:44]:44s} icache req {req:.3e} miss {miss:.3e} ({100 * miss / max(req, 1):.1f} %) dup {c.get('SQC_ICACHE_MISSES_DUPLICATE', 0):.3e}  ifetch {c.get('SQ_IFETCH', 0):.3e} level/fetch {c.get('SQ_IFETCH_LEVEL', 0) / max(c.get('SQ_IFETCH', 1), 1):.1f}  wave_cycles {c.get('SQ_WAVE_CYCLES', 0):.3e}")
PY
