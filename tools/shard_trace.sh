#!/bin/bash
# usage (GPU box): tools/shard_trace.sh TAG [WORLD=8] -> gpurun_out/TAG_shard_trace.txt: per launch (last render of each run), the whole
# frame's duration / WORLD against rank 0's share rendered alone: where the strong-scaling loss of one rank sits, launch by launch.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r3}; W=${2:-8}
for w in 1 $W; do
  WORLD=$w rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_st_$w -o run -- python3 $R/tools/shard_trace.py > $R/gpurun_out/${TAG}_st_$w.log 2>&1 || { echo "run $w failed"; tail -3 $R/gpurun_out/${TAG}_st_$w.log; exit 1; }
done
python3 - <<PY > $R/gpurun_out/${TAG}_shard_trace.txt
import csv, glob, sys
sys.path.insert(0, "$R")
from bench import kernel_base
def launches(w):
    f = glob.glob("$R/gpurun_out/${TAG}_st_%d/**/*kernel_trace.csv" % w, recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if (kernel_base(r["Kernel_Name"]) or "").startswith(("k_wf_", "k_gmon", "k_sampler"))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last render = everything from the last k_sampler_tables launch on
    last = max(i for i, r in enumerate(rows) if kernel_base(r["Kernel_Name"]) == "k_sampler_tables")
    return [(kernel_base(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6, int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows[last:]]
a, b = launches(1), launches($W)
assert [k for k, *_ in a] == [k for k, *_ in b], "launch sequences differ"
print("# launch by launch: whole frame / $W  vs  rank 0 of $W rendered alone (ms); gap = idle time on the stream before the launch (us)")
print("%-26s %10s %10s %8s %10s" % ("kernel", "whole/$W", "rank 0", "ratio", "gap us"))
tot_a = tot_b = gap = 0.0
bounce = -1
for i, ((k, ta, *_), (_, tb, sb, eb)) in enumerate(zip(a, b)):
    g = (sb - b[i - 1][3]) * 1e-3 if i else 0.0
    tot_a += ta / $W; tot_b += tb; gap += g
    if ta / $W > 0.05 or tb > 0.05:
        print("%-26s %10.3f %10.3f %8.2f %10.1f" % (k, ta / $W, tb, tb / (ta / $W) if ta else 0, g))
print("%-26s %10.3f %10.3f %8.3f %10.1f  (%d launches)" % ("sum of kernels", tot_a, tot_b, tot_b / tot_a, gap, len(a)))
print("span of rank 0's render: %.3f ms" % ((b[-1][3] - b[0][2]) * 1e-6))
PY
cat $R/gpurun_out/${TAG}_shard_trace.txt
