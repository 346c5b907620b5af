#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc counters: python tools/pmc_summary.py <dir> [kernel-name substring ...]"""
import collections, csv, glob, json, sys
subs = sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(f"{sys.argv[1]}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if subs and not any(x in name for x in subs):
            continue
        key = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:70]
        a = acc[key][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k, cs in acc.items():
    print(json.dumps({"kernel": k, **{c: round(v[0] / max(v[1], 1), 1) for c, v in sorted(cs.items())}, "dispatches": max(v[1] for v in cs.values())}))
