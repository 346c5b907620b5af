#!/bin/bash
# Same-box A/B of bench.py flag sets, alternating runs (boxes differ by up to 5 %, so only runs of ONE gpurun call are compared).
# Usage (on the GPU box): bash tools/bench_ab.sh OUTDIR ROUNDS STEPS "flags of A" "flags of B" ["flags of C" ...]
set -e
out=$1; rounds=$2; steps=$3; shift 3
mkdir -p "$out"
: > "$out/bench_ab.jsonl"
for r in $(seq 1 "$rounds"); do
  for v in "$@"; do
    python bench.py --steps "$steps" --warmup 3 --no-cpu-baseline $v | tail -1 > "$out/line.json"
    python - "$out/line.json" "$v" >> "$out/bench_ab.jsonl" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps(dict(flags=sys.argv[2], value=round(d["value"], 1), ms_per_step=round(d["ms_per_step"], 3), sclk_mhz=round(d["clock"]["sclk_mhz_mean"]), first_ms=d["step_done_ms"][0], steady_ms=round((d["step_done_ms"][-2] - d["step_done_ms"][1]) / (len(d["step_done_ms"]) - 3), 3), last_ms=round(d["step_done_ms"][-1] - d["step_done_ms"][-2], 2))))
PY
    tail -1 "$out/bench_ab.jsonl"
  done
done
