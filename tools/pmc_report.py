"""Summarise rocprofv3 --pmc results per kernel (SQ counters). Accepts output directories that
hold either *counter_collection.csv or the rocpd sqlite database (run_results.db)."""
import collections, csv, glob, sqlite3, sys
names = ["k_wf_compact_commit", "k_wf_compact", "k_wf_extend_retry_lean", "k_wf_shadow_retry_lean", "k_wf_extend_lean", "k_wf_shadow_lean", "k_wf_extend_fast", "k_wf_shadow_fast", "k_wf_shadow", "k_sampler_tables", "k_wf_trace_extend", "k_wf_trace_shadow", "k_wf_extend", "k_wf_connect", "k_wf_shade", "k_wf_roulette",
         "k_wf_generate", "k_gmon_blend", "k_render_mega"]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
info, dur = {}, collections.defaultdict(float)
for path in sys.argv[1:]:
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((x for x in names if x in r["Kernel_Name"]), None)
            if k:
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                info[k] = (r["VGPR_Count"], r["LDS_Block_Size"], r["Scratch_Size"])
    for f in glob.glob(path + "/**/*.db", recursive=True):
        db = sqlite3.connect(f)
        for kn, cn, v, vg, lds, scr in db.execute(
                "select kernel_name, counter_name, value, vgpr_count, lds_block_size, scratch_size from counters_collection"):
            k = next((x for x in names if x in kn), None)
            if k:
                agg[k][cn] += float(v)
                info[k] = (vg, lds, scr)
        if not dur:
            for kn, d in db.execute("select name, duration from kernels"):
                k = next((x for x in names if x in kn), None)
                if k:
                    dur[k] += d * 1e-6
for k, v in agg.items():
    print(k, "vgpr/lds/scratch", info[k], "total ms %.3f" % dur.get(k, 0))
    for c, x in sorted(v.items()):
        print(f"   {c:28s} {x:.4e}")
    if "SQ_WAVE_CYCLES" in v and "SQ_ACTIVE_INST_VALU" in v:
        wc = v["SQ_WAVE_CYCLES"]
        print("   lane_util = %.3f  active_valu/wave_cycles = %.3f  wait_inst_any/wave_cycles = %.3f  busy_cycles %.3e" % (
            v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64), v["SQ_ACTIVE_INST_VALU"] / wc,
            v.get("SQ_WAIT_INST_ANY", 0) / wc, v.get("SQ_BUSY_CYCLES", 0)))
    if "SQ_INST_LEVEL_VMEM" in v and v.get("SQ_INSTS_VMEM_RD"):
        print("   vmem latency (level/insts) = %.1f cycles" % (v["SQ_INST_LEVEL_VMEM"] / v["SQ_INSTS_VMEM_RD"]))
