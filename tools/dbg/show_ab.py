#!/usr/bin/env python3
"""Print the median microseconds (and the difference against the first selector) of a tools/gemm_bench.py result file."""
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f'{d["label"]:24s} {d["shape"]:20s}', {k[3:]: (round(v["us_med"], 1), v["diff_vs_first"]) for k, v in d.items() if k.startswith("cfg")})
