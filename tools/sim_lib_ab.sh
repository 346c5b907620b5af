#!/bin/bash
# Same-box A/B of two BUILDS of libcor_amd.so on tools/sim_bench.py (one shape), alternating processes inside ONE gpurun call.
# Usage (GPU box): bash tools/sim_lib_ab.sh OUT.jsonl OTHER.so SHAPE [ROUNDS]     SHAPE: 1m | 512x125000 | 512x12500 ...
out=$1; other=$2; shape=$3; rounds=${4:-3}
: > "$out"
for r in $(seq 1 "$rounds"); do
  for lib in "$other" ""; do
    COR_AMD_LIB=$lib python3 tools/sim_bench.py "$shape" 12 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print(json.dumps(dict(lib=sys.argv[1] or 'this build', Bq=d['Bq'], Ng=d['Ng'], us_back_to_back=round(d['us_back_to_back'], 1), us_single=round(d['us_single_median'], 1), frac=round(d['roofline']['frac'], 4))))" "$lib" | tee -a "$out"
  done
done
