#!/bin/bash
# Per-kernel average durations of cor_similarity_topk at one shape for two builds (rocprofv3 --kernel-trace --stats). Usage: sim_kernels.sh SHAPE [OTHER.so]
shape=$1; other=$2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for lib in "$other" ""; do
  tag=$( [ -z "$lib" ] && echo this || echo other )
  rm -rf $R/gpurun_out/prof_simk_$tag
  COR_AMD_LIB=$( [ -z "$lib" ] && echo "" || echo $R/$lib ) timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_simk_$tag -- python3 $R/tools/sim_bench.py $shape 20 > /dev/null 2>&1
  f=$(ls $R/gpurun_out/prof_simk_$tag/*/*kernel_stats.csv | head -1)
  python3 - "$f" "$tag" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "sim_" in r["Name"]:
        print(sys.argv[2], r["Name"].split("(")[1][20:] if False else r["Name"][27:75], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done
