#!/bin/bash
# Same-box A/B of two BUILDS of libcor_amd.so on the bench (driver's form), alternating runs inside ONE gpurun call.
# Usage (GPU box): bash tools/bench_lib_ab.sh OUT.jsonl OTHER.so [ROUNDS]      ("" = this tree's build)
out=$1; other=$2; rounds=${3:-3}
: > "$out"
for r in $(seq 1 "$rounds"); do
  for lib in "$other" ""; do
    COR_AMD_LIB=$lib python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
print(json.dumps(dict(lib=sys.argv[1] or 'this build', value=round(d['value'], 1), ms_per_step=round(d['ms_per_step'], 3), gemm_frac=round(d['roofline']['frac'], 4), sclk_mhz=round(d['clock']['sclk_mhz_mean']))))" "$lib" | tee -a "$out"
  done
done
