"""Summarise a scripts/collect_pmc.sh run: kernel stats CSV + per-kernel mean FETCH_SIZE / WRITE_SIZE + durations."""
import collections, csv, glob, json, shutil, sys
out, tag = sys.argv[1], sys.argv[2]
line = json.loads(open("%s/bench_line_under_rocprof.json" % out).read().strip().splitlines()[-1])
cfg = line["config"]
summary = {"tag": tag,
           "workload": {"n": cfg["n_codes"], "queries": cfg["queries_per_step"], "topk": cfg["topk"], "gpus": line["n_gpus"],
                        "m": 8 if "m=8" in line["metric"] else 16, "data": "pipeline" if "DeltaTree" in cfg["workload"] else "stream",
                        "description": cfg["workload"]},
           "unit": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch (rocprofv3); bytes = value * 1024",
           "gfx950_correction": "FETCH_SIZE reads 1/2 of a wide coalesced stream (MI355X_MICROARCH.md HBM section): "
                                "hbm_read_bytes = 2 * FETCH_SIZE * 1024; dword-per-lane loads are uncalibrated",
           "kernels": {}}
short = lambda n: n.split("(")[0].replace("void ", "")
for name in ("fetch", "write"):
    f = glob.glob("%s/pmc_%s/*/*_counter_collection.csv" % (out, name))[0]
    agg = collections.defaultdict(list)
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and "End_Timestamp" in r:
            dur[short(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for (k, c), v in agg.items():
        summary["kernels"].setdefault(k, {})[c] = {"dispatches": len(v), "mean_kib": sum(v) / len(v), "max_kib": max(v)}
    for k, v in dur.items():
        summary["kernels"].setdefault(k, {})["avg_ns_under_pmc_" + name] = sum(v) / len(v)
stats = glob.glob("%s/stats/*/*_kernel_stats.csv" % out)[0]
shutil.copy(stats, "%s/%s_kernel_stats.csv" % (out, tag))
for r in csv.DictReader(open(stats)):
    summary["kernels"].setdefault(short(r["Name"]), {})["stats"] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                                                                    "total_ns": int(r["TotalDurationNs"]), "pct": float(r["Percentage"])}
# the scan instantiation this workload runs most (<8, true, ...> = plain codes from the per-batch decode scratch;
# <8, false, ...> = decode inside the scan, e.g. the 64-query parity gate)
scans = [(v.get("stats", {}).get("calls", 0), k, v) for k, v in summary["kernels"].items() if k.startswith("dpq::scan_kernel")]
sc = max(scans)[2] if scans else {}
if scans:
    summary["scan_kernel_instantiation"] = max(scans)[1]
dk = next((v for k, v in summary["kernels"].items() if k.startswith("dpq::decode_list_kernel")), {})
if "FETCH_SIZE" in dk and "WRITE_SIZE" in dk:
    summary["decode_list_kernel_hbm_bytes_per_launch"] = 2 * dk["FETCH_SIZE"]["mean_kib"] * 1024 + dk["WRITE_SIZE"]["mean_kib"] * 1024
    summary["decode_list_kernel_avg_launch_ms_kernel_trace"] = dk["stats"]["avg_ns"] / 1e6
if "FETCH_SIZE" in sc and "WRITE_SIZE" in sc:
    summary["scan_kernel_hbm_bytes_per_launch"] = 2 * sc["FETCH_SIZE"]["mean_kib"] * 1024 + sc["WRITE_SIZE"]["mean_kib"] * 1024
    summary["scan_kernel_hbm_bytes_per_launch_uncorrected"] = (sc["FETCH_SIZE"]["mean_kib"] + sc["WRITE_SIZE"]["mean_kib"]) * 1024
    summary["scan_kernel_avg_launch_ms_under_pmc"] = sc.get("avg_ns_under_pmc_fetch", sc["stats"]["avg_ns"]) / 1e6
    summary["scan_kernel_avg_launch_ms_kernel_trace"] = sc["stats"]["avg_ns"] / 1e6
    summary["scan_kernel_hbm_GBps"] = summary["scan_kernel_hbm_bytes_per_launch"] / (summary["scan_kernel_avg_launch_ms_kernel_trace"] * 1e-3) / 1e9
# the stream pass: strand_kernel (lane per run, M = 8 with a bootstrap) or stream_kernel (wavefront per chunk)
def first(prefixes):
    for pre in prefixes:
        for k, v in summary["kernels"].items():
            if k.startswith(pre):
                return k, v
    return None, {}
# (one query per pass: strand1_kernel; several: strand_kernel; small shards / M = 16: stream_kernel)
summary["stream_pass_kernel"], st = first(("dpq::strand1_kernel", "dpq::strand_kernel", "dpq::stream_kernel"))
if "FETCH_SIZE" in st and "WRITE_SIZE" in st:   # one query per pass
    summary["stream_kernel_hbm_bytes_per_launch"] = 2 * st["FETCH_SIZE"]["mean_kib"] * 1024 + st["WRITE_SIZE"]["mean_kib"] * 1024
    summary["stream_kernel_avg_launch_ms_kernel_trace"] = st["stats"]["avg_ns"] / 1e6
    summary["stream_kernel_hbm_GBps"] = summary["stream_kernel_hbm_bytes_per_launch"] / (st["stats"]["avg_ns"] * 1e-9) / 1e9
summary["bench_line_under_rocprof"] = {k: line[k] for k in ("value", "ms_per_step", "repetitions") if k in line}
json.dump(summary, open("%s/%s_pmc_summary.json" % (out, tag), "w"), indent=1)
print(json.dumps({k: summary[k] for k in summary if k.startswith("scan_kernel") or k.startswith("decode_list") or k.startswith("stream_kernel")}, indent=1))
for k, v in summary["kernels"].items():
    if "stats" in v and k.startswith("dpq::") and "anonymous" not in k:
        print("%-46s calls %5d avg %9.1f us" % (k[:46], v["stats"]["calls"], v["stats"]["avg_ns"] / 1e3))
