#!/usr/bin/env python3
"""Print the similarity measurement pass of tools/archive/measure_r04_sim.sh (gpurun_out/r4s): back-to-back times per path and per-kernel averages."""
import csv, glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r4s"
rows = {}
for f in ("sim_bench", "sim_bench_global", "sim_bench_global_wavefinal"):
    p = os.path.join(d, f + ".jsonl")
    if not os.path.exists(p):
        continue
    for l in open(p):
        j = json.loads(l)
        rows.setdefault((j["Bq"], j["Ng"], j["dtype"][6:]), {})[f] = (j["us_back_to_back"], j["roofline"]["frac"])
print("%-22s %22s %22s %22s" % ("shape", "default", "global thr (block final)", "global thr (wave final)"))
for k, v in rows.items():
    print("%-22s" % ("%dx%d %s" % k), *["%12.1f us %6.3f" % v[f] if f in v else " " * 22 for f in ("sim_bench", "sim_bench_global", "sim_bench_global_wavefinal")])
for pd in sorted(glob.glob(os.path.join(d, "prof_*"))):
    fs = sorted(glob.glob(os.path.join(pd, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
    if not fs:
        continue
    print("==", os.path.basename(pd))
    for r in csv.DictReader(open(fs[-1])):
        if "sim_" in r["Name"]:
            print("  %-62s calls %3s avg %8.1f us min %8.1f" % (r["Name"].replace("(anonymous namespace)::", "")[:62], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
