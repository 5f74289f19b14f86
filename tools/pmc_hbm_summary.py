"""profiles/r1_pmc_hbm.json from the two --pmc passes of tools/pmc_hbm.sh (gpurun_out/pmc_hbm_{rd,wr})."""
import csv, collections, json, os, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 bench.py --steps 1 --warmup 0 "
                  "--no-cpu-baseline --no-roofline (separate passes; tools/pmc_hbm.sh)",
       "unit_note": "FETCH_SIZE / WRITE_SIZE are KiB at the L2's memory side (Infinity-Cache hits included); FETCH_SIZE is "
                    "doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B read requests at 64 B); WRITE_SIZE as reported",
       "workload": "sponza_class 1920x1080, 256 spp, 8 bounces (bench.py default), one step", "kernels": {}}
for tag, cn in (("rd", "FETCH_SIZE"), ("wr", "WRITE_SIZE")):
    src = os.path.join(ROOT, "gpurun_out", f"pmc_hbm_{tag}", "run_counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(src)):
        if r["Counter_Name"] != cn or "k_" not in r["Kernel_Name"]:
            continue
        n = r["Kernel_Name"]
        k = n[n.find("k_"):].split("(")[0]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    for k, (c, v) in agg.items():
        e = out["kernels"].setdefault(k, {})
        e["launches"] = c
        if cn == "FETCH_SIZE":
            e["read_bytes_per_launch"] = int(v / c * 1024 * 2)
        else:
            e["write_bytes_per_launch"] = int(v / c * 1024)
    shutil.copy(src, os.path.join(ROOT, "profiles", f"r1_pmc_{cn.lower()}_counter_collection.csv"))
json.dump(out, open(os.path.join(ROOT, "profiles", "r1_pmc_hbm.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    if v.get("read_bytes_per_launch", 0) > 1e8:
        print(k, v)
