#!/bin/bash
# Throughput of the benchmarked path over the per-GPU batch and the number of forwards in flight (serving view): bench.py --batch B
# --inflight F, hipGraph replay, one GPU. Output: gpurun_out/sweep/batch_sweep.jsonl -> profiles/r05_batch_sweep.jsonl
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sweep; mkdir -p $O; : > $O/batch_sweep.jsonl
cd $R
for B in 1 2 4 8 16 32 64; do
  S=$((B <= 4 ? 60 : (B <= 16 ? 30 : 20)))
  for F in 1 2 3; do
    timeout -k 10 200 python3 bench.py --batch $B --inflight $F --steps $S --warmup 3 --no-cpu-baseline 2>/dev/null | grep '^{"metric"' >> $O/batch_sweep.jsonl || { echo "FAILED batch $B inflight $F"; exit 1; }
  done
  echo "batch $B done"
done
python3 - <<'PY'
import json, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
for l in open(f"{R}/gpurun_out/sweep/batch_sweep.jsonl"):
    d = json.loads(l)
    print(d["config"]["global_batch"], d["config"]["forwards_in_flight"], round(d["value"], 1), "triplets/s", round(d["ms_per_step"], 2), "ms/step", "first result", d["step_done_ms"][0], "ms")
PY
