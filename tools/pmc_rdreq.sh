#!/bin/bash
# What do FETCH_SIZE's kilobytes mean for a GATHER kernel? (MI355X_MICROARCH.md: "other access widths are uncalibrated")
# One pass over one un-timed step of the bench workload with the L2's memory-side read requests split BY SIZE
# (TCC_EA0_RDREQ_{32B,64B,128B}: exact bytes = 32 n32 + 64 n64 + 128 n128), one with FETCH_SIZE, per kernel.
# usage (GPU box): tools/pmc_rdreq.sh TAG [bench args]  -> gpurun_out/TAG_rdreq.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r3}; shift
for pass in "sizes:TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctrs=${pass#*:}
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_rdreq_$name -o run -- python3 $R/bench.py --pmc-child "$@" > $R/gpurun_out/${TAG}_rdreq_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $R/gpurun_out/${TAG}_rdreq_$name.log; }
done
python3 - <<PY > $R/gpurun_out/${TAG}_rdreq.txt
import csv, glob, collections, sys
sys.path.insert(0, "$R")
from bench import kernel_base
def load(name):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob("$R/gpurun_out/${TAG}_rdreq_%s/**/*counter_collection.csv" % name, recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = kernel_base(r["Kernel_Name"])
            if k is None: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (r.get("Dispatch_Id"), k)
            if key not in seen: seen.add(key); n[k] += 1
    return agg, n
sz, n = load("sizes"); fs, _ = load("fetch"); ws, _ = load("write")
print("# per kernel, summed over ONE step of the bench workload (bytes in GB): exact = 32 n32 + 64 n64 + 128 n128 of TCC_EA0_RDREQ_*")
print("%-26s %8s %10s %10s %10s %10s %12s %12s %8s %10s" % ("kernel", "launches", "n32 (M)", "n64 (M)", "n128 (M)", "rdreq (M)", "exact GB", "FETCH_SIZE GB", "ratio", "WRITE GB"))
for k in sorted(sz, key=lambda k: -sz[k].get("TCC_EA0_RDREQ_sum", 0)):
    a = sz[k]; n32, n64, n128 = (a.get("TCC_EA0_RDREQ_%s_sum" % s, 0.0) for s in ("32B", "64B", "128B")); tot = a.get("TCC_EA0_RDREQ_sum", 0.0)
    exact = 32 * n32 + 64 * n64 + 128 * n128
    other = tot - n32 - n64 - n128
    f = fs.get(k, {}).get("FETCH_SIZE", 0.0) * 1024
    w = ws.get(k, {}).get("WRITE_SIZE", 0.0) * 1024
    print("%-26s %8d %10.1f %10.1f %10.1f %10.1f %12.2f %12.2f %8.3f %10.2f%s" % (k, n[k], n32 / 1e6, n64 / 1e6, n128 / 1e6, tot / 1e6, exact / 1e9, f / 1e9, exact / f if f else 0, w / 1e9,
          "  (%.1f M requests of no listed size)" % (other / 1e6) if abs(other) > 0.01 * max(tot, 1) else ""))
PY
cat $R/gpurun_out/${TAG}_rdreq.txt
