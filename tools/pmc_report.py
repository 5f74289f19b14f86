"""Summarise a rocprofv3 --pmc counter_collection.csv per kernel (SQ counters)."""
import collections, csv, glob, sys
path = sys.argv[1]
files = glob.glob(path + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float))
info = {}
names = ["k_wf_trace_extend", "k_wf_trace_shadow", "k_wf_extend", "k_wf_connect", "k_wf_shade", "k_wf_post",
         "k_wf_generate", "k_gmon_blend", "k_render_mega"]
for f in files:
    for r in csv.DictReader(open(f)):
        k = next((x for x in names if x in r["Kernel_Name"]), None)
        if not k:
            continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        info[k] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
for k, v in agg.items():
    print(k, "vgpr/lds/scratch", info[k])
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:.4e}")
    if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_VALU" in v:
        wc = v["SQ_WAVE_CYCLES"]
        print("   lane_util = %.3f  active_valu/wave_cycles = %.3f  wait_inst_any/wave_cycles = %.3f" % (
            v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64), v["SQ_ACTIVE_INST_VALU"] / wc,
            v.get("SQ_WAIT_INST_ANY", 0) / wc))
