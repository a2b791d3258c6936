"""Summarise a scripts/collect_pmc.sh run: kernel stats CSV + per-kernel mean FETCH_SIZE / WRITE_SIZE."""
import collections, csv, glob, json, shutil, sys
out, tag = sys.argv[1], sys.argv[2]
summary = {"tag": tag, "unit": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (rocprofv3); bytes = value * 1024",
           "gfx950_correction": "FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section): "
                                "hbm_read_bytes = 2 * FETCH_SIZE * 1024; dword-per-lane loads are uncalibrated",
           "kernels": {}}
for name in ("fetch", "write"):
    f = glob.glob("%s/pmc_%s/*/*_counter_collection.csv" % (out, name))[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in agg.items():
        summary["kernels"].setdefault(k, {})[c] = {"dispatches": len(v), "mean_kib": sum(v) / len(v), "max_kib": max(v)}
stats = glob.glob("%s/stats/*/*_kernel_stats.csv" % out)[0]
shutil.copy(stats, "%s/%s_kernel_stats.csv" % (out, tag))
for r in csv.DictReader(open(stats)):
    k = r["Name"].split("(")[0].replace("void ", "")
    summary["kernels"].setdefault(k, {})["stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                     "total_ns": int(r["TotalDurationNs"]), "pct": float(r["Percentage"])}
sc = next((v for k, v in summary["kernels"].items() if k.startswith("dpq::scan_kernel")), {})
if "FETCH_SIZE" in sc and "WRITE_SIZE" in sc:
    summary["scan_kernel_hbm_bytes_per_launch"] = 2 * sc["FETCH_SIZE"]["mean_kib"] * 1024 + sc["WRITE_SIZE"]["mean_kib"] * 1024
    summary["scan_kernel_hbm_bytes_per_launch_uncorrected"] = (sc["FETCH_SIZE"]["mean_kib"] + sc["WRITE_SIZE"]["mean_kib"]) * 1024
json.dump(summary, open("%s/%s_pmc_summary.json" % (out, tag), "w"), indent=1)
print(json.dumps(summary, indent=1)[:1500])
