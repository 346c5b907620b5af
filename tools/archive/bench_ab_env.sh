#!/bin/bash
# as tools/bench_ab.sh, but the variants are ENVIRONMENT assignments followed by bench.py flags, e.g. "GPU_MAX_HW_QUEUES=8 --inflight 2"
# usage (GPU box): bash tools/bench_ab_env.sh OUT ROUNDS STEPS "VAR=x [flags]" ...
set -e
out=$1; rounds=$2; steps=$3; shift 3; VARS=("$@")
mkdir -p "$out"
: > "$out/bench_ab_env.jsonl"
for r in $(seq 1 "$rounds"); do
  for v in "${VARS[@]}"; do
    e=${v%% *}; f=""; [ "$v" != "$e" ] && f=${v#* }
    env $e python bench.py --steps "$steps" --warmup 3 --no-cpu-baseline $f | tail -1 > "$out/line.json"
    python - "$out/line.json" "$v" >> "$out/bench_ab_env.jsonl" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(json.dumps(dict(variant=sys.argv[2], value=round(d["value"], 1), ms_per_step=round(d["ms_per_step"], 3), first_ms=d["step_done_ms"][0], steady_ms=round((d["step_done_ms"][-2] - d["step_done_ms"][1]) / (len(d["step_done_ms"]) - 3), 3))))
PY
    tail -1 "$out/bench_ab_env.jsonl"
  done
done
