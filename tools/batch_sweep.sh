#!/bin/bash
# Throughput / latency of the benchmarked path over the per-GPU batch (serving view): bench.py --batch B, hipGraph replay, one GPU.
# Output: gpurun_out/sweep/batch_sweep.jsonl (one bench line per batch) -> profiles/r03_batch_sweep.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sweep; mkdir -p $O; : > $O/batch_sweep.jsonl
cd $R
for B in 1 2 4 8 16 32 64; do
  S=$((B <= 4 ? 20 : (B <= 16 ? 10 : 5)))
  timeout -k 10 200 python3 bench.py --batch $B --steps $S --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' >> $O/batch_sweep.jsonl || { echo "FAILED batch $B"; exit 1; }
  echo "batch $B done"
done
python3 - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for l in open(f"{R}/gpurun_out/sweep/batch_sweep.jsonl"):
    d = json.loads(l)
    print(d["config"]["global_batch"], round(d["value"], 1), "triplets/s", round(d["ms_per_step"], 2), "ms/step", "gemm frac", round(d["roofline"]["frac"], 3))
PY
